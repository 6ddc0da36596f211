"""The production transport path (lane-per-column kernels, isopycnal coefficients
folded once per step; uvic2.9_amd/csrc/kernels_col.hpp) against the oracle, the
reference's golden runs and the bit-exact GPU path.  Folding re-associates fp64
products, so the bar is a stated tolerance, not equality:
  one step : |diff| <= 1e-13 * max|field|   per tracer
  20 steps : |diff| <= 1e-12 * max|field|
(100 steps against the compiled reference: tests/test_gpu_drift100.py)"""
from pathlib import Path

import numpy as np
import pytest

from uvic29_amd import synthetic, performance_set
import oracle_c

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).resolve().parent / "golden"


def _rel(a, b):
    return np.abs(a - b).max() / np.abs(b).max()


def _rand_src(oc, seed=2029, scale=1e-9):
    g = oc.grid
    rng = np.random.default_rng(seed)
    return np.asfortranarray(rng.standard_normal((g.imt, g.km, g.jmt, oc.cfg.nsrc)) * scale * oc.topo.tmask[..., None])


@pytest.mark.parametrize("cfg,dims", [("p2", (14, 14, 6)), ("c30", (14, 14, 6)), ("c30", (102, 102, 19)),
                                      ("p2", (202, 202, 32)), ("c30", (202, 202, 32)), ("perf15", (102, 102, 19)),
                                      # grids that are not square, not multiples of the wave or of anything else
                                      ("c30", (23, 17, 6)), ("c30", (17, 39, 19)), ("p2", (71, 14, 6)), ("c30", (38, 101, 19))])
@pytest.mark.parametrize("mode", [False, "columns"])
def test_one_step_vs_oracle(cfg, dims, mode):
    """mode False: production (T, S through the bit-exact kernels, the others through the column kernels); "columns":
    every tracer through the column kernels"""
    from uvic29_amd.tracer import TracerModel
    oc = synthetic.make_ocean(performance_set(int(cfg[4:])) if cfg.startswith("perf") else cfg, *dims)
    to, so, c = synthetic.load_eos(dims[2])
    src = _rand_src(oc) if oc.cfg.nsrc else None
    orc = oracle_c.Oracle(oc, to=to, so=so, c=c, src=src)
    orc.isopyc(); orc.add_k33()
    want = orc.transport()
    m = TracerModel(*dims, oc.cfg.nt, oc.cfg.nsrc, oc.cfg.ntnpzd)
    m.set_exact(mode)
    m.load_ocean(oc, to, so, c, src=src)
    m.isopyc(); m.transport(); m.convect()
    got = m.download("t_taup1")
    jmt = dims[1]
    if mode is False:   # T and S of the production path are the reference's bits
        assert np.array_equal(got[:, :, 1:jmt - 1, :2], want[:, :, 1:jmt - 1, :2])
    worst = 0.0
    for n, name in enumerate(oc.cfg.tracers):
        r = _rel(got[:, :, 1:jmt - 1, n], want[:, :, 1:jmt - 1, n])
        worst = max(worst, r)
        assert r <= 1e-13, (name, r)
    # cyclic images are written by the kernels
    assert np.array_equal(got[0, :, 1:jmt - 1], got[dims[0] - 2, :, 1:jmt - 1])
    assert np.array_equal(got[dims[0] - 1, :, 1:jmt - 1], got[1, :, 1:jmt - 1])
    print("worst one-step relative difference", worst)
    m.close()


@pytest.mark.parametrize("cfg", ["p2", "c30"])
def test_twenty_steps_vs_reference_golden_run(cfg):
    from uvic29_amd.tracer import TracerModel, TimeLoop
    oc = synthetic.make_ocean(cfg, 14, 14, 6)
    to, so, c = synthetic.load_eos(6)
    g = np.load(GOLD / f"run_{cfg}_14x14x6_n20.npz")
    m = TracerModel(14, 14, 6, oc.cfg.nt, oc.cfg.nsrc, oc.cfg.ntnpzd)
    m.load_ocean(oc, to, so, c)
    if oc.cfg.ntnpzd:
        m.set_mobi(oc)
    loop = TimeLoop(m, oc.params.dtts, oc.params.nmix)
    for _ in range(20):
        loop.step()
    m.sync()
    got = m.download("t_tau")
    worst = max(_rel(got[:, :, 1:13, n], g["t"][:, :, 1:13, n]) for n in range(oc.cfg.nt))
    print("worst relative drift after 20 steps", worst)
    assert worst <= 1e-12
    m.close()


@pytest.mark.parametrize("mode", [False, "columns"])
@pytest.mark.parametrize("dims", [(14, 14, 6), (102, 102, 19)])
def test_whole_step_with_fused_convection_equals_the_separate_passes(dims, mode):
    """`uvic_gpu_tracer` sends T and S through both column passes and the convective walk first (side stream) and lets
    pass B of the other tracers replay the mixed segments before t(tau+1) is stored; `transport` + `convect` run pass A
    and B for all tracers together and convct2 as its own two passes.  Same bits, and convection does act on this ocean."""
    from uvic29_amd.tracer import TracerModel
    oc = synthetic.make_ocean("c30", *dims)
    to, so, c = synthetic.load_eos(dims[2])
    src = _rand_src(oc)
    out = {}
    for how in ("fused", "separate", "unmixed"):
        m = TracerModel(*dims, oc.cfg.nt, oc.cfg.nsrc, 0)      # sources given: the comparison is about transport + convection
        m.set_exact(mode)
        m.load_ocean(oc, to, so, c, src=src)
        m.isopyc()
        if how == "fused":
            m.tracer()
        else:
            m.transport()
            if how == "separate":
                m.convect()
        out[how] = m.download("t_taup1")
        m.close()
    assert np.array_equal(out["fused"], out["separate"])
    changed = (out["separate"] != out["unmixed"]).any(axis=(0, 1, 2))
    assert changed[:2].all() and changed[2:].any(), "convection left the step unchanged: the test ocean has no unstable column"
