"""MOBI source terms: oracle vs compiled reference / golden fixtures (CPU), the
kernel source under host emulation vs oracle (CPU, bit-exact because both use
libm), and the GPU kernel vs oracle (tolerance: device exp/log/pow/tanh differ
from libm in the last bits)."""
import ctypes
from pathlib import Path

import numpy as np
import pytest

from uvic29_amd import synthetic, mobi as pm
import mobi_c
import oracle_c
import refmodel

GOLD = Path(__file__).resolve().parent / "golden"


def test_mobi_param_fixture_matches_reference_init():
    if not refmodel.available("c30", 14, 14, 6):
        pytest.skip("oracle/_ref not built")
    import refdriver
    oc = synthetic.make_ocean("c30", 14, 14, 6)
    ro = refdriver.RefOcean(oc)
    prm = pm.load_table("c30", 6)
    for n, val in prm.items():
        assert np.array_equal(np.atleast_1d(val), ro.v[n].reshape(-1)), n


def test_oracle_step_with_mobi_matches_golden_c30():
    """isopyc + MOBI sources + transport + convection of the oracle == reference `tracer`."""
    oc = synthetic.make_ocean("c30", 14, 14, 6)
    to, so, c = synthetic.load_eos(6)
    prm = pm.load_table("c30", 6)
    c2 = 2 * oc.params.dtts
    src = mobi_c.mobi_sources(oc, prm, oc.t_taum1, c2)
    orc = oracle_c.Oracle(oc, to=to, so=so, c=c, src=src)
    orc.isopyc(); orc.add_k33()
    tp = orc.transport()
    g = np.load(GOLD / "step_c30_14x14x6.npz")
    assert np.array_equal(tp[:, :, 1:13], g["t_taup1"][:, :, 1:13])


def test_mobi_driver_columns_match_compiled_reference():
    if not refmodel.available("c30", 14, 14, 6):
        pytest.skip("oracle/_ref not built")
    import refdriver
    oc = synthetic.make_ocean("c30", 14, 14, 6)
    cfg, km = oc.cfg, 6
    ro = refdriver.RefOcean(oc)
    prm = pm.load_table("c30", km)
    c2 = 2 * oc.params.dtts
    P = mobi_c.make_params(cfg, oc.grid, prm, c2)
    for n in ("nbio", "dtbio", "rdtts", "rnbio"):
        ro.ref.set(n, getattr(P, n))
    lib = oracle_c.lib()
    rng = np.random.default_rng(7)
    d = ctypes.c_double
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    checked = 0
    for trial in range(60):
        i, j = rng.integers(1, 13), rng.integers(1, 13)
        kmx = int(oc.topo.kmt[i, j])
        if kmx == 0:
            continue
        tn = np.zeros((km, cfg.ntnpzd), order="F")
        for m, name in enumerate(cfg.mobi):
            tn[:, m] = oc.t_taum1[i, :, j, cfg.index(name) - 1]
        if trial % 3 == 0:      # collisions with trcmin: negative-prevention flags
            tn[rng.integers(0, km), rng.integers(0, cfg.ntnpzd)] = 1e-13
        t_in = oc.t_taum1[i, :, j, 0].copy()
        o2 = oc.t_taum1[i, :, j, cfg.index("o2") - 1] * 1000. * (0.02 if trial % 5 == 0 else 1.0)  # suboxic columns
        s_in = 1e3 * oc.t_taum1[i, :, j, 1] + 35.
        aou = 200.0 - o2
        dic = oc.t_taum1[i, :, j, cfg.index("dic") - 1].copy()
        alk = oc.t_taum1[i, :, j, cfg.index("alk") - 1].copy()
        sgb = np.zeros(km); sgb[kmx - 1] = 1.0; sgb[max(kmx - 2, 0)] = 0.3
        tn1, tn2 = tn.copy(order="F"), tn.copy(order="F")
        s1, s2 = np.zeros((km, cfg.nsrc), order="F"), np.zeros((km, cfg.nsrc), order="F")
        ro.ref.call("mobi_driver", kmx, c2, 5e-4, 0.45, 90.0, tn1, t_in, o2, aou, s_in, dic, alk, 280.0, sgb, s1)
        lib.orc_mobi_driver(ctypes.byref(P), kmx, d(c2), d(5e-4), d(0.45), d(90.0), p(tn2), p(t_in), p(o2), p(aou),
                            p(s_in), p(dic), p(alk), d(280.0), p(sgb), p(s2))
        assert np.array_equal(s1, s2) and np.array_equal(tn1, tn2)
        checked += 1
    assert checked > 20


@pytest.mark.parametrize("dims", [(14, 14, 6)])
def test_mobi_kernel_source_under_host_emulation_equals_oracle(dims):
    import emu
    oc = synthetic.make_ocean("c30", *dims)
    prm = pm.load_table("c30", dims[2])
    to, so, c = synthetic.load_eos(dims[2])
    src_o = mobi_c.mobi_sources(oc, prm, oc.t_taum1, 2 * oc.params.dtts)
    em = emu.EmuOcean(oc, to, so, c)
    assert np.array_equal(em.mobi(prm), src_o)
    # the four-wave team form (threads + barrier stand in for the workgroup): same bits
    em.a["src"][...] = -1.0
    got = em.mobi(prm, team=True)
    assert np.array_equal(got[1:-1, :, 1:-1], src_o[1:-1, :, 1:-1])   # columns 2..imt-1, rows 2..jmt-1 are computed
    # the carbonate solve with shared reciprocals (what the device runs by default): same Newton path, values to rounding
    em.a["src"][...] = -1.0
    got = em.mobi(prm, carb_shared=True)
    for s in range(src_o.shape[-1]):
        scale = np.abs(src_o[..., s]).max()
        assert np.abs(got[1:-1, :, 1:-1, s] - src_o[1:-1, :, 1:-1, s]).max() <= 1e-13 * scale, s


MOBI_RTOL = 1e-11  # per source slot, relative to the slot max; measured on MI355X: 4.7e-13 (102x102x19), 5.8e-15 (14x14x6)


@pytest.mark.gpu
@pytest.mark.parametrize("dims", [(14, 14, 6), (102, 102, 19), (23, 17, 6), (38, 101, 19)])   # (the last two: not square, odd)
def test_gpu_mobi_sources_vs_oracle(dims):
    from uvic29_amd.tracer import TracerModel
    oc = synthetic.make_ocean("c30", *dims)
    prm = pm.load_table("c30", dims[2])
    to, so, c = synthetic.load_eos(dims[2])
    want = mobi_c.mobi_sources(oc, prm, oc.t_taum1, 2 * oc.params.dtts)
    m = TracerModel(*dims, oc.cfg.nt, oc.cfg.nsrc, oc.cfg.ntnpzd)
    m.load_ocean(oc, to, so, c)
    m.set_mobi(oc)
    m.mobi()
    got = m.download("src")
    worst = 0.0
    for s, name in enumerate(oc.cfg.sources):
        scale = np.abs(want[..., s]).max()
        err = np.abs(got[..., s] - want[..., s]).max()
        worst = max(worst, err / scale)
        assert err <= MOBI_RTOL * scale, (name, err, scale)
    print("worst relative source error", worst)
    m.close()


@pytest.mark.gpu
def test_gpu_full_step_with_mobi_vs_golden_and_oracle():
    """One complete `tracer` step (MOBI + transport + convection) on the GPU against the
    reference's golden output; transport amplifies the MOBI rounding differences by 2*dt."""
    from uvic29_amd.tracer import TracerModel
    oc = synthetic.make_ocean("c30", 14, 14, 6)
    to, so, c = synthetic.load_eos(6)
    g = np.load(GOLD / "step_c30_14x14x6.npz")
    m = TracerModel(14, 14, 6, oc.cfg.nt, oc.cfg.nsrc, oc.cfg.ntnpzd)
    m.set_exact(True)
    m.load_ocean(oc, to, so, c)
    m.set_mobi(oc)
    m.isopyc(); m.tracer()
    got = m.download("t_taup1")
    for n, name in enumerate(oc.cfg.tracers):
        a, b = got[:, :, 1:13, n], g["t_taup1"][:, :, 1:13, n]
        assert np.abs(a - b).max() <= 1e-11 * np.abs(b).max(), (name, np.abs(a - b).max(), np.abs(b).max())
    # T and S have no biological source: still bit-exact
    assert np.array_equal(got[:, :, 1:13, :2], g["t_taup1"][:, :, 1:13, :2])
    m.close()


@pytest.mark.gpu
def test_gpu_twenty_steps_with_mobi_drift_vs_reference_run():
    """20 steps (one mixing step) device-resident vs the reference's golden run: the
    north-star drift criterion is < 1e-12 relative after 100 steps for tracers on the
    same arithmetic; the device exp/log/pow differ from libm in the last bits, measured drift 2.9e-15."""
    from uvic29_amd.tracer import TracerModel
    oc = synthetic.make_ocean("c30", 14, 14, 6)
    to, so, c = synthetic.load_eos(6)
    g = np.load(GOLD / "run_c30_14x14x6_n20.npz")
    m = TracerModel(14, 14, 6, oc.cfg.nt, oc.cfg.nsrc, oc.cfg.ntnpzd)
    m.set_exact(True)
    m.load_ocean(oc, to, so, c)
    m.set_mobi(oc)
    dtts, nmix = oc.params.dtts, oc.params.nmix
    for it in range(1, 21):
        mixing = (it % nmix) == 0
        if mixing:
            m.upload("t_taum1", m.download("t_tau"))
        m.set_params(c2dtts=dtts if mixing else 2.0 * dtts)
        m.isopyc(); m.tracer(); m.rotate()
    got = m.download("t_tau")
    worst = 0.0
    for n, name in enumerate(oc.cfg.tracers):
        a, b = got[:, :, 1:13, n], g["t"][:, :, 1:13, n]
        rel = np.abs(a - b).max() / np.abs(b).max()
        worst = max(worst, rel)
        assert rel <= 1e-12, (name, rel)   # measured on MI355X: 2.9e-15
    print("worst relative drift after 20 steps", worst)
    m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("prefetch", [False, True])
def test_gpu_time_loop_with_source_prefetch_matches_golden_run(prefetch):
    """The device-resident schedule (mixing step by buffer aliasing, MOBI sources one step
    ahead on a side stream) reproduces the reference's 20-step run."""
    from uvic29_amd.tracer import TracerModel, TimeLoop
    oc = synthetic.make_ocean("c30", 14, 14, 6)
    to, so, c = synthetic.load_eos(6)
    g = np.load(GOLD / "run_c30_14x14x6_n20.npz")
    m = TracerModel(14, 14, 6, oc.cfg.nt, oc.cfg.nsrc, oc.cfg.ntnpzd)
    m.set_exact(True)
    m.load_ocean(oc, to, so, c)
    m.set_mobi(oc)
    loop = TimeLoop(m, oc.params.dtts, oc.params.nmix, prefetch=prefetch)
    for _ in range(20):
        loop.step()
    m.sync()
    got = m.download("t_tau")
    for n, name in enumerate(oc.cfg.tracers):
        a, b = got[:, :, 1:13, n], g["t"][:, :, 1:13, n]
        assert np.abs(a - b).max() <= 1e-12 * np.abs(b).max(), (name, prefetch)
    assert np.array_equal(got[:, :, 1:13, :2], g["t"][:, :, 1:13, :2])
    m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("exact", [True, False])
def test_lookahead_with_an_advancing_clock_equals_inline(exact):
    """The reference advances relyr every ocean step and takes the month of the dust field and the solar declination
    from it (u09/mom/tracer.F:311-338).  The look-ahead chain of step n+1 runs during step n: it must compute with the
    clock of step n+1 (uvic_gpu_step_lookahead_at), or a run with look-ahead differs from one that computes its sources
    in line.  Six steps across a month boundary (relyr 1/12), both ways: same bits."""
    from uvic29_amd.tracer import TracerModel, TimeLoop
    oc = synthetic.make_ocean("c30", 14, 14, 6)
    to, so, c = synthetic.load_eos(6)
    dyr = oc.params.dtts / (365.0 * 86400.0)
    clock = (1.0 / 12.0 - 2.5 * dyr, dyr, oc.forcing.co2ccn)      # the boundary falls between steps 3 and 4
    out = {}
    for ahead in (True, False):
        m = TracerModel(14, 14, 6, oc.cfg.nt, oc.cfg.nsrc, oc.cfg.ntnpzd)
        m.set_exact(exact)
        m.load_ocean(oc, to, so, c)
        m.set_mobi(oc)
        loop = TimeLoop(m, oc.params.dtts, oc.params.nmix, prefetch=ahead, clock=clock)
        for _ in range(6):
            loop.step()
        m.sync()
        out[ahead] = m.download("t_tau")
        m.close()
    assert np.array_equal(out[True], out[False])
    # ... and the clock matters: a run whose clock stands still gives other sources
    m = TracerModel(14, 14, 6, oc.cfg.nt, oc.cfg.nsrc, oc.cfg.ntnpzd)
    m.set_exact(exact)
    m.load_ocean(oc, to, so, c)
    m.set_mobi(oc)
    loop = TimeLoop(m, oc.params.dtts, oc.params.nmix)
    for _ in range(6):
        loop.step()
    m.sync()
    still = m.download("t_tau")
    m.close()
    assert not np.array_equal(still, out[True])
