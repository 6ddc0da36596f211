"""Parity of the HIP path (through the C ABI) with the oracle, the committed
golden fixtures and -- when oracle/_ref travelled with the tree -- the compiled
reference itself.  Transport is integer-free fp64 arithmetic evaluated in the
reference's order without FMA contraction on both sides, so the bar is
BIT-EXACT equality (tolerance 0) for isopyc, FCT advection, isoflux, the
explicit update, the tridiagonal solve and convection."""
from pathlib import Path

import numpy as np
import pytest

from uvic29_amd import synthetic, performance_set
import oracle_c
import refmodel

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).resolve().parent / "golden"


def _model(oc, to, so, c, src=None):
    from uvic29_amd.tracer import TracerModel
    g = oc.grid
    m = TracerModel(g.imt, g.jmt, g.km, oc.cfg.nt, oc.cfg.nsrc, oc.cfg.ntnpzd, device=0)
    m.set_exact(True)        # this file checks the bit-exact formulation (kernels_fct.hpp)
    m.load_ocean(oc, to, so, c, src=src)
    return m


def _rand_src(oc, seed=2029, scale=1e-9):
    g = oc.grid
    rng = np.random.default_rng(seed)
    s = rng.standard_normal((g.imt, g.km, g.jmt, oc.cfg.nsrc)) * scale
    return np.asfortranarray(s * oc.topo.tmask[..., None])


ISO = ("alphai", "betai", "ddxt", "ddyt", "ddzt", "Ai_ez", "Ai_nz", "Ai_bx", "Ai_by", "K11", "K22", "K33",
       "adv_vetiso", "adv_vbtiso")


@pytest.mark.parametrize("cfg,dims", [("p2", (14, 14, 6)), ("c30", (14, 14, 6)), ("p2", (102, 102, 19)),
                                      ("c30", (23, 17, 6)), ("p2", (17, 39, 19)), ("c30", (71, 14, 6))])   # (odd, non-square grids)
def test_isopyc_transport_convect_bit_exact_vs_oracle(cfg, dims):
    oc = synthetic.make_ocean(cfg, *dims)
    to, so, c = synthetic.load_eos(dims[2])
    src = _rand_src(oc) if oc.cfg.nsrc else None
    orc = oracle_c.Oracle(oc, to=to, so=so, c=c, src=src)
    m = _model(oc, to, so, c, src=src)
    orc.isopyc(); orc.add_k33()
    m.isopyc()
    for n in ISO:
        assert np.array_equal(m.download(n), orc.a[n]), n
    imt, jmt = dims[0], dims[1]
    assert np.array_equal(m.download("adv_vntiso")[:imt - 1], orc.a["adv_vntiso"][:imt - 1])
    assert np.array_equal(m.download("diff_cbt")[1:imt - 1, :, 1:jmt - 1], orc.a["diff_cbt"][1:imt - 1, :, 1:jmt - 1])
    want = orc.transport()
    m.transport()
    m.convect()
    got = m.download("t_taup1")
    assert np.array_equal(got[:, :, 1:jmt - 1], want[:, :, 1:jmt - 1])
    m.close()


def test_full_size_nt30_bit_exact_vs_oracle():
    """BASELINE config 4 shape (102x102x19, nt=30) with a random source term."""
    oc = synthetic.make_ocean("c30")
    to, so, c = synthetic.load_eos(19)
    src = _rand_src(oc)
    orc = oracle_c.Oracle(oc, to=to, so=so, c=c, src=src)
    m = _model(oc, to, so, c, src=src)
    orc.isopyc(); orc.add_k33(); m.isopyc()
    want = orc.transport()
    m.transport(); m.convect()
    got = m.download("t_taup1")
    assert np.array_equal(got[:, :, 1:101], want[:, :, 1:101])
    m.close()


def test_step_matches_golden_fixture_p2():
    oc = synthetic.make_ocean("p2", 14, 14, 6)
    to, so, c = synthetic.load_eos(6)
    g = np.load(GOLD / "step_p2_14x14x6.npz")
    m = _model(oc, to, so, c)
    m.isopyc()
    m.tracer()
    got = m.download("t_taup1")
    assert np.array_equal(got[:, :, 1:13], g["t_taup1"][:, :, 1:13])
    assert np.array_equal(m.download("k33")[:, :, 1:13], g["k33"])
    m.close()


def test_twenty_steps_match_golden_run_p2():
    """20 leapfrog steps with a forward (mixing) step every nmix-th, device resident."""
    oc = synthetic.make_ocean("p2", 14, 14, 6)
    to, so, c = synthetic.load_eos(6)
    g = np.load(GOLD / "run_p2_14x14x6_n20.npz")
    m = _model(oc, to, so, c)
    dtts, nmix = oc.params.dtts, oc.params.nmix
    for it in range(1, 21):
        mixing = (it % nmix) == 0
        if mixing:
            m.upload("t_taum1", m.download("t_tau"))
        m.set_params(c2dtts=dtts if mixing else 2.0 * dtts)
        m.isopyc(); m.tracer(); m.rotate()
    got = m.download("t_tau")
    assert np.array_equal(got[:, :, 1:13], g["t"][:, :, 1:13])
    m.close()


@pytest.mark.parametrize("nchunk", [1, 3])
def test_longitude_chunking_is_invisible(nchunk):
    oc = synthetic.make_ocean(performance_set(8), 102, 102, 19)
    to, so, c = synthetic.load_eos(19)
    src = _rand_src(oc)
    orc = oracle_c.Oracle(oc, to=to, so=so, c=c, src=src)
    m = _model(oc, to, so, c, src=src)
    m.set_option("nchunk", nchunk)
    orc.isopyc(); orc.add_k33(); m.isopyc()
    want = orc.transport()
    m.transport(); m.convect()
    assert np.array_equal(m.download("t_taup1")[:, :, 1:101], want[:, :, 1:101])
    m.close()


def test_refined_grid_202x202x32():
    """BASELINE config 5 shape: the tile no longer fits LDS in one piece."""
    oc = synthetic.make_ocean("p2", 202, 202, 32)
    to, so, c = synthetic.load_eos(32)
    orc = oracle_c.Oracle(oc, to=to, so=so, c=c)
    m = _model(oc, to, so, c)
    orc.isopyc(); orc.add_k33(); m.isopyc()
    want = orc.transport()
    m.transport(); m.convect()
    assert np.array_equal(m.download("t_taup1")[:, :, 1:201], want[:, :, 1:201])
    m.close()


def test_against_compiled_reference_when_present():
    if not refmodel.available("p2", 102, 102, 19):
        pytest.skip("oracle/_ref did not travel with the tree")
    import refdriver
    oc = synthetic.make_ocean("p2")
    ro = refdriver.RefOcean(oc)
    tp = ro.step()
    m = _model(oc, ro.v["to"], ro.v["so"], ro.v["c"])
    m.isopyc(); m.tracer()
    assert np.array_equal(m.download("t_taup1")[:, :, 1:101], tp[:, :, 1:101])
    m.close()


def test_error_behaviour():
    from uvic29_amd.tracer import TracerModel, UvicGpuError
    m = TracerModel(14, 14, 6, 2)
    m.tracer(js=5, je=4)            # `if (js .gt. je) return`, tracer.F:219
    with pytest.raises(UvicGpuError):
        m.tracer(joff=3)
    with pytest.raises(UvicGpuError):
        m.upload("t_tau", np.zeros((3, 3)))
    with pytest.raises(UvicGpuError):
        m.transport()               # c2dtts not set
    m.close()


def test_twenty_steps_time_loop_with_lookahead_match_golden_run_p2():
    """The same 20 steps through TimeLoop, where the isopyc products of step n+1 are computed one step ahead on a
    side stream into an alternate buffer set (uvic_gpu_prefetch_isopyc): bit-identical with the exact kernels."""
    from uvic29_amd.tracer import TimeLoop
    oc = synthetic.make_ocean("p2", 14, 14, 6)
    to, so, c = synthetic.load_eos(6)
    g = np.load(GOLD / "run_p2_14x14x6_n20.npz")
    m = _model(oc, to, so, c)
    loop = TimeLoop(m, oc.params.dtts, oc.params.nmix)
    for _ in range(20):
        loop.step()
    m.sync()
    got = m.download("t_tau")
    assert np.array_equal(got[:, :, 1:13], g["t"][:, :, 1:13])
    m.close()
