"""Baroclinic momentum step: `state` + `clinic` with the U-cell part of `adv_vel`, the
bottom drag of `setvbc`, `isbcu`/`asbcu` and the polar filter `filuv`.

CPU: C restatement == compiled reference (configuration "m2" of oracle/build_ref.py), bit for bit; the same against
     the committed fixture tests/golden/clinic_m2_14x14x6.npz (outputs of the compiled reference); host-emulated
     kernels == C restatement, bit for bit.
GPU: library == C restatement, bit for bit (integer-exact order of operations, no contraction), at 14x14x6 and at
     BASELINE's 102x102x19; size-independent properties at 102x102x19 (no vertical mean left, land untouched,
     cyclic columns)."""
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
for p in (ROOT, ROOT / "oracle", ROOT / "tests" / "hostemu"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

GRIDS = [(14, 14, 6), (102, 102, 19)]


def _setup(imt, jmt, km):
    from uvic29_amd import synthetic
    oc = synthetic.make_ocean("m2", imt, jmt, km)
    mom = synthetic.make_momentum(oc.grid, oc.topo, oc.u)
    return oc, mom, synthetic.load_eos(km), synthetic.make_filter_u(oc.grid, km)


def _oracle_step(oc, mom, eos, flt=None):
    import oracle_c
    rho = oracle_c.state(oc.grid, eos, oc.t_tau[..., 0], oc.t_tau[..., 1])
    M = oracle_c.Momentum(oc, mom, rho)
    up, zu = M.step()
    if flt is not None:
        up = oracle_c.filuv(up, oc.grid, oc.topo, mom, flt)
    return rho, M, up, zu


@pytest.mark.parametrize("imt,jmt,km", GRIDS)
@pytest.mark.parametrize("filtered", [False, True])
def test_oracle_equals_reference(imt, jmt, km, filtered):
    import oracle_c
    import refdriver
    import refmodel
    if not refmodel.available("m2", imt, jmt, km):
        pytest.skip("oracle/_ref build m2 %dx%dx%d not present" % (imt, jmt, km))
    oc, mom, eos, flt = _setup(imt, jmt, km)
    R = refdriver.RefOcean(oc)
    R.set_momentum(mom)
    if filtered:
        R.set_filter_u(flt)
    rho = oracle_c.state(oc.grid, eos, oc.t_tau[..., 0], oc.t_tau[..., 1])
    assert np.array_equal(rho, R.state())
    M = oracle_c.Momentum(oc, mom, rho)
    for got, want in zip(M.adv_vel_u(), R.adv_vel_u()):
        assert np.array_equal(got, want)
    smf, bmf = R.setvbc()
    assert np.array_equal(smf, mom.smf) and np.array_equal(M.bmf(), bmf)
    up_ref, zu_ref, gp_ref = R.clinic()
    up, zu = M.clinic()
    if filtered:
        up = oracle_c.filuv(up, oc.grid, oc.topo, mom, flt)
    assert np.array_equal(M.a["grad_p"], gp_ref)
    assert np.array_equal(zu, zu_ref)
    assert np.array_equal(up, up_ref)
    assert np.abs(up).max() > 0.1 and np.abs(zu).max() > 1e-5


def test_oracle_sbc_accumulation_equals_reference():
    import oracle_c
    import refdriver
    import refmodel
    if not refmodel.available("m2", 14, 14, 6):
        pytest.skip("oracle/_ref build m2 14x14x6 not present")
    oc, mom, eos, _ = _setup(14, 14, 6)
    R = refdriver.RefOcean(oc)
    R.set_momentum(mom)
    M = oracle_c.Momentum(oc, mom, R.state())
    S, v = R.ref.set, R.v
    S("igu", 11); S("igv", 12); S("isu", 13); S("isv", 14); S("ntspos", 3)
    rng = np.random.default_rng(0)
    for p in (10, 11, 12, 13):
        v["sbc"][:, :, p] = rng.standard_normal((14, 14))
    for osegs, osege in ((1, 0), (0, 0), (0, 1), (1, 1)):
        S("osegs", osegs); S("osege", osege)
        planes = [np.array(v["sbc"][:, :, p], order="F") for p in (10, 11, 12, 13)]
        R.clinic()
        M.sbcu("i", planes[0], planes[1], osegs, osege, 1.0 / 3)
        M.sbcu("a", planes[2], planes[3], osegs, osege, 1.0 / 3)
        for q in range(4):
            assert np.array_equal(planes[q], v["sbc"][:, :, 10 + q])


def test_oracle_matches_golden_fixture():
    """The pinning without the compiled reference at hand (GPU box): the fixture holds the reference's own outputs
    (tests/golden/make_golden.py clinic_fixture)."""
    gold = np.load(ROOT / "tests" / "golden" / "clinic_m2_14x14x6.npz")
    oc, mom, eos, flt = _setup(14, 14, 6)
    rho, M, up, zu = _oracle_step(oc, mom, eos)
    assert np.array_equal(rho, gold["rho"])
    assert np.array_equal(M.a["grad_p"], gold["grad_p"])
    assert np.array_equal(zu, gold["zu"]) and np.array_equal(up, gold["u_taup1"])
    _, _, upf, _ = _oracle_step(oc, mom, eos, flt)
    assert np.array_equal(upf, gold["u_taup1_filtered"])
    assert not np.array_equal(upf, up)


@pytest.mark.parametrize("imt,jmt,km", GRIDS)
def test_hostemu_equals_oracle(imt, jmt, km):
    import emu
    oc, mom, eos, flt = _setup(imt, jmt, km)
    rho, M, up, zu = _oracle_step(oc, mom, eos)
    E = emu.EmuMomentum(oc, mom, eos)
    assert np.array_equal(E.state()[:, :, 1:], rho[:, :, 1:])
    got_u, got_zu = E.clinic()
    assert np.array_equal(E.a["grad_p"], M.a["grad_p"])
    assert np.array_equal(got_zu, zu) and np.array_equal(got_u, up)
    _, _, upf, _ = _oracle_step(oc, mom, eos, flt)
    assert np.array_equal(E.filuv(oc.grid, flt), upf) and not np.array_equal(upf, up)
    # the surface-velocity accumulators, all four phases of a coupling segment
    rng = np.random.default_rng(1)
    planes = [np.asfortranarray(rng.standard_normal((imt, jmt))) for _ in range(4)]
    for q, n in enumerate(("sbc_gu", "sbc_gv", "sbc_su", "sbc_sv")):
        E.a[n][...] = planes[q]
    for osegs, osege in ((1, 0), (0, 0), (0, 1), (1, 1)):
        M.sbcu("i", planes[0], planes[1], osegs, osege, 0.25)
        M.sbcu("a", planes[2], planes[3], osegs, osege, 0.25)
        E.sbcu(osegs | (osege << 1), 0.25)
        for q, n in enumerate(("sbc_gu", "sbc_gv", "sbc_su", "sbc_sv")):
            assert np.array_equal(E.a[n], planes[q])


def _gpu_model(oc, mom, eos):
    from uvic29_amd.tracer import TracerModel
    g = oc.grid
    m = TracerModel(g.imt, g.jmt, g.km, oc.cfg.nt, oc.cfg.nsrc, oc.cfg.ntnpzd, device=0)
    m.load_ocean(oc, *eos)
    m.load_momentum(oc, mom)
    return m


@pytest.mark.gpu
@pytest.mark.parametrize("imt,jmt,km", GRIDS)
def test_gpu_state_and_clinic_equal_oracle(imt, jmt, km):
    oc, mom, eos, flt = _setup(imt, jmt, km)
    rho, M, up, zu = _oracle_step(oc, mom, eos)
    m = _gpu_model(oc, mom, eos)
    m.state()
    assert np.array_equal(m.download("rho")[:, :, 1:], rho[:, :, 1:])
    got_u, got_zu = m.clinic()
    assert np.array_equal(m.download("grad_p"), M.a["grad_p"])
    assert np.array_equal(got_zu, zu)
    assert np.array_equal(got_u, up)
    # with the polar filter
    _, _, upf, _ = _oracle_step(oc, mom, eos, flt)
    m.set_filter_u(oc, flt)
    got_u, got_zu = m.clinic()
    assert np.array_equal(got_zu, zu) and np.array_equal(got_u, upf)
    m.close()


@pytest.mark.gpu
def test_gpu_clinic_on_the_refined_grid_and_in_two_slabs():
    """BASELINE config 5's grid (202x202x32): state + clinic bit-identical to the oracle, computed whole and as two
    latitude slabs (uvic_gpu_set_shard js..je: each call fills its own U rows, the inputs around them are present)."""
    oc, mom, eos, flt = _setup(202, 202, 32)
    rho, M, up, zu = _oracle_step(oc, mom, eos, flt)
    m = _gpu_model(oc, mom, eos)
    m.set_filter_u(oc, flt)
    m.state()
    got_u, got_zu = m.clinic()
    assert np.array_equal(got_zu, zu) and np.array_equal(got_u, up)
    for n in ("up1", "up2"):
        m.upload(n, np.zeros(m.shape(n), order="F"))
    m.upload("zu", np.zeros(m.shape("zu"), order="F"))
    for js, je in ((2, 101), (102, 201)):
        m.set_shard(js=js, je=je)
        m.clinic_only()
    m.set_shard(js=2, je=201)
    got_u = np.stack([m.download("up1"), m.download("up2")], axis=-1)
    assert np.array_equal(m.download("zu"), zu) and np.array_equal(got_u, up)
    m.close()


@pytest.mark.gpu
def test_gpu_clinic_matches_golden_fixture():
    gold = np.load(ROOT / "tests" / "golden" / "clinic_m2_14x14x6.npz")
    oc, mom, eos, flt = _setup(14, 14, 6)
    m = _gpu_model(oc, mom, eos)
    m.state()
    got_u, got_zu = m.clinic()
    assert np.array_equal(got_zu, gold["zu"]) and np.array_equal(got_u, gold["u_taup1"])
    m.set_filter_u(oc, flt)
    got_u, _ = m.clinic()
    assert np.array_equal(got_u, gold["u_taup1_filtered"])
    m.close()


@pytest.mark.gpu
def test_gpu_sbc_accumulation_equals_oracle():
    oc, mom, eos, _ = _setup(14, 14, 6)
    rho, M, up, zu = _oracle_step(oc, mom, eos)
    m = _gpu_model(oc, mom, eos)
    m.state()
    rng = np.random.default_rng(1)
    names = ("sbc_gu", "sbc_gv", "sbc_su", "sbc_sv")
    planes = [np.asfortranarray(rng.standard_normal((14, 14))) for _ in range(4)]
    for q, n in enumerate(names):
        m.upload(n, planes[q])
    for osegs, osege in ((1, 0), (0, 0), (0, 1), (1, 1)):
        M.sbcu("i", planes[0], planes[1], osegs, osege, 0.25)
        M.sbcu("a", planes[2], planes[3], osegs, osege, 0.25)
        m.clinic(accumulate_sbc=True, osegs=bool(osegs), osege=bool(osege), rts=0.25)
        for q, n in enumerate(names):
            assert np.array_equal(m.download(n), planes[q])
    m.close()


@pytest.mark.gpu
def test_gpu_clinic_properties_at_full_size():
    """Size-independent properties on BASELINE's grid: internal modes have no vertical mean, land stays at rest,
    the cyclic columns are images, zu is the depth average of the tendency."""
    oc, mom, eos, flt = _setup(102, 102, 19)
    g, topo = oc.grid, oc.topo
    m = _gpu_model(oc, mom, eos)
    m.set_filter_u(oc, flt)
    m.state()
    u, zu = m.clinic()
    for n in range(2):
        mean = np.einsum("ikj,k->ij", u[..., n], g.dzt) * mom.hr
        scale = np.abs(u[..., n]).max()
        assert np.abs(mean[1:-1, 1:-1]).max() <= 1e-12 * scale
        assert np.all(u[..., n][topo.umask == 0.0][...] == 0.0) or np.all((u[..., n] * (1 - topo.umask))[1:-1, :, 1:-1] == 0.0)
        assert np.array_equal(u[0, :, 1:-1, n], u[-2, :, 1:-1, n]) and np.array_equal(u[-1, :, 1:-1, n], u[1, :, 1:-1, n])
    assert np.all(zu[topo.kmu == 0] == 0.0)
    m.close()
