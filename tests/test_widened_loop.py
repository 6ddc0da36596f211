"""The widened device-resident sequence of one ocean step -- adv_vel, isopyc, vmixc (tidal mixing + K33),
tracer with the polar filter (mom.F:332-389) -- over several steps against the same sequence of the compiled
reference routines driven through their COMMON blocks."""
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
for p in (ROOT, ROOT / "oracle"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

pytestmark = pytest.mark.gpu
NSTEP = 8


@pytest.mark.parametrize("cfg,dims", [("p2", (14, 14, 6)), ("p2", (102, 102, 19))])
def test_device_sequence_matches_reference_sequence(cfg, dims):
    import refmodel
    if not refmodel.available(cfg, *dims):
        pytest.skip("oracle/_ref did not travel with the tree")
    import refdriver
    from uvic29_amd import synthetic
    from uvic29_amd.tracer import TracerModel
    imt, jmt, km = dims
    oc = synthetic.make_ocean(cfg, imt, jmt, km)
    tid = synthetic.make_tidal(oc.grid, oc.topo, oc.params.kappa_h)
    flt = synthetic.make_filter(oc.grid, km)
    ro = refdriver.RefOcean(oc)
    if not hasattr(ro.ref.lib, "findex_"):
        pytest.skip("oracle/_ref predates findex")
    ro.set_filter(flt)
    prev = np.asfortranarray(oc.diff_cbt_bg)          # what vmixc finds below the bottom level on the first step

    m = TracerModel(imt, jmt, km, oc.cfg.nt, oc.cfg.nsrc, oc.cfg.ntnpzd, device=0)
    m.load_ocean(oc, ro.v["to"], ro.v["so"], ro.v["c"])
    m.set_exact(True)
    m.load_velocity(oc)
    m.load_tidal(oc, tid)
    m.set_filter(oc, flt)
    m.set_params(diff_cbt_has_k33=1)
    m.upload("diff_cbt", prev)
    for n in ("adv_vet", "adv_vnt", "adv_vbt"):
        m.upload(n, np.zeros(m.shape(n), order="F"))
    ro.v["diff_cbt"][...] = prev[:, :, 1:jmt - 1]
    for n in ("edrm2", "edrs2", "edrk1", "edro1"):             # COMMON /tdr/ (tidal_kv.h) of the reference
        ro.v[n][...] = getattr(tid, n)
    for n in ("zetar", "ogamma", "gravrho0r", "kappa_h"):
        ro.ref.set(n, getattr(tid, n))

    for step in range(NSTEP):
        ro.ref.call("adv_vel", 0, 1, jmt, 2, imt - 1)
        ro.isopyc()
        ro.ref.call("vmixc", 0, 1, jmt, 2, imt - 1)
        ro.tracer()
        m.adv_vel(); m.isopyc(); m.vmixc(); m.tracer()
        want = ro.v["t"][..., 2]
        got = m.download("t_taup1")
        # exact kernels, and vmixc's exponentials from the host's table: every step bit for bit
        assert np.array_equal(got[:, :, 1:-1], want[:, :, 1:-1]), (step, np.abs(got - want)[:, :, 1:-1].max())
        ro.rotate(); m.rotate()
    assert np.isfinite(got).all()
    m.close()
