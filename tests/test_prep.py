"""Producers of the tracer step's shared inputs: `adv_vel` and the tracer
part of `vmixc` (tidal mixing + K33).

CPU: C restatement == compiled reference (bit for bit), host-emulated kernels == C restatement.
GPU: library == C restatement, bit for bit (vmixc's exponentials come from a table made with the host's exp)."""
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
for p in (ROOT, ROOT / "oracle", ROOT / "tests" / "hostemu"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

GRIDS = [("p2", 14, 14, 6), ("c30", 102, 102, 19)]


def _setup(cfg, imt, jmt, km):
    from uvic29_amd import synthetic
    oc = synthetic.make_ocean(cfg, imt, jmt, km)
    tid = synthetic.make_tidal(oc.grid, oc.topo, oc.params.kappa_h)
    return oc, tid, synthetic.load_eos(km)


def _oracle_vmixc(oc, tid, eos):
    import oracle_c
    to, so, c = eos
    orc = oracle_c.Oracle(oc, to=to, so=so, c=c)
    orc.isopyc()
    prev = np.asfortranarray(oc.diff_cbt_bg)
    return oracle_c.vmixc(oc.grid, oc.topo, tid, orc.a["alphai"], orc.a["betai"], orc.a["ddzt"], orc.a["K33"], prev), prev


@pytest.mark.parametrize("cfg,imt,jmt,km", GRIDS)
def test_oracle_equals_reference(cfg, imt, jmt, km):
    import oracle_c
    import refmodel
    if not refmodel.available(cfg, imt, jmt, km):
        pytest.skip("oracle/_ref not built for this grid")
    import refdriver
    oc, tid, eos = _setup(cfg, imt, jmt, km)
    ro = refdriver.RefOcean(oc)
    rvet, rvnt, rvbt = ro.adv_vel()
    vet, vnt, vbt = oracle_c.adv_vel(oc.grid, oc.u)
    assert np.array_equal(vet[:, :, 1:], rvet[:, :, 1:]) and np.array_equal(vnt, rvnt) and np.array_equal(vbt[:, :, 1:], rvbt[:, :, 1:])
    ro.isopyc()
    want, prev = _oracle_vmixc(oc, tid, eos)
    got = ro.vmixc(tid, prev)
    assert np.array_equal(want[1:-1, :, 1:-1], got[1:-1, :, 1:-1])
    assert (got[1:-1, :, 1:-1] > 1.5 * oc.params.kappa_h).mean() > 0.3      # the tidal term is active


@pytest.mark.parametrize("cfg,imt,jmt,km", GRIDS)
def test_hostemu_equals_oracle(cfg, imt, jmt, km):
    import emu
    import oracle_c
    oc, tid, eos = _setup(cfg, imt, jmt, km)
    em = emu.EmuOcean(oc, *eos)
    vet, vnt, vbt = oracle_c.adv_vel(oc.grid, oc.u)
    evet, evnt, evbt = em.adv_vel()
    assert np.array_equal(evet[:, :, 1:], vet[:, :, 1:]) and np.array_equal(evnt, vnt) and np.array_equal(evbt[:, :, 1:], vbt[:, :, 1:])
    want, prev = _oracle_vmixc(oc, tid, eos)
    em.isopyc()
    got = em.vmixc(tid, prev)
    assert np.array_equal(want[1:-1, :, 1:-1], got[1:-1, :, 1:-1])


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,imt,jmt,km", GRIDS)
def test_gpu_adv_vel_and_vmixc(cfg, imt, jmt, km):
    import oracle_c
    from uvic29_amd.tracer import TracerModel
    oc, tid, eos = _setup(cfg, imt, jmt, km)
    m = TracerModel(imt, jmt, km, oc.cfg.nt, oc.cfg.nsrc, oc.cfg.ntnpzd, device=0)
    m.load_ocean(oc, *eos)
    for n in ("adv_vet", "adv_vnt", "adv_vbt"):          # make sure the kernels produce them
        m.upload(n, np.zeros(m.shape(n), order="F"))
    m.load_velocity(oc)
    m.adv_vel()
    vet, vnt, vbt = oracle_c.adv_vel(oc.grid, oc.u)
    assert np.array_equal(m.download("adv_vet")[:, :, 1:], vet[:, :, 1:])
    assert np.array_equal(m.download("adv_vnt"), vnt)
    assert np.array_equal(m.download("adv_vbt")[:, :, 1:], vbt[:, :, 1:])
    want, prev = _oracle_vmixc(oc, tid, eos)
    m.load_tidal(oc, tid)
    m.set_params(diff_cbt_has_k33=1)
    m.upload("diff_cbt", prev)
    m.isopyc()
    m.vmixc()
    got = m.download("diff_cbt")
    # bit for bit: the two exponentials of the tidal term are tabulated on the host with the reference's own exp
    assert np.array_equal(got[1:-1, :, 1:-1], want[1:-1, :, 1:-1])
    # a whole step with the device-made inputs equals the step with the uploaded ones
    m.close()


def test_oracle_matches_golden_fixture():
    """The same pinning without the compiled reference at hand: tests/golden/prep_p2_14x14x6.npz holds the
    reference's own outputs (tests/golden/make_golden.py)."""
    import oracle_c
    from uvic29_amd import synthetic
    gold = np.load(ROOT / "tests" / "golden" / "prep_p2_14x14x6.npz")
    oc, tid, eos = _setup("p2", 14, 14, 6)
    vet, vnt, vbt = oracle_c.adv_vel(oc.grid, oc.u)
    assert np.array_equal(vet[:, :, 1:], gold["adv_vet"][:, :, 1:]) and np.array_equal(vnt, gold["adv_vnt"])
    assert np.array_equal(vbt[:, :, 1:], gold["adv_vbt"][:, :, 1:])
    want, _ = _oracle_vmixc(oc, tid, eos)
    assert np.array_equal(want[1:-1, :, 1:-1], gold["diff_cbt"][1:-1, :, 1:-1])
    flt = synthetic.make_filter(oc.grid, 6)
    istf, ietf = oracle_c.findex(oc.topo.kmt, flt)
    assert np.array_equal(istf, gold["istf"]) and np.array_equal(ietf, gold["ietf"])
    to, so, c = eos
    orc = oracle_c.Oracle(oc, to=to, so=so, c=c)
    orc.isopyc(); orc.add_k33()
    got = oracle_c.setbcx(oracle_c.filt(np.array(orc.transport(), order="F"), oc.grid, oc.topo, flt, istf, ietf))
    assert np.array_equal(got[:, :, 1:-1], gold["t_taup1_filtered"][:, :, 1:-1])
