"""The CPU restatement (oracle/uvic_oracle.c) against (a) the compiled reference
in oracle/_ref when it is present (build container) and (b) the committed golden
fixtures generated from it (always)."""
from pathlib import Path

import numpy as np
import pytest

from uvic29_amd import synthetic
import oracle_c
import refmodel

GOLD = Path(__file__).resolve().parent / "golden"


def _oracle_step(oc):
    to, so, c = synthetic.load_eos(oc.grid.km)
    orc = oracle_c.Oracle(oc, to=to, so=so, c=c)
    orc.isopyc()
    orc.add_k33()
    return orc


def test_isopyc_and_transport_match_golden_p2():
    oc = synthetic.make_ocean("p2", 14, 14, 6)
    g = np.load(GOLD / "step_p2_14x14x6.npz")
    orc = _oracle_step(oc)
    jmt, imt = 14, 14
    assert np.array_equal(orc.a["K33"][:, :, 1:jmt - 1], g["k33"])
    assert np.array_equal(orc.a["adv_vetiso"][:, :, 1:jmt - 1], g["adv_vetiso"])
    assert np.array_equal(orc.a["adv_vntiso"][:imt - 1, :, :jmt - 1], g["adv_vntiso"][:imt - 1])
    assert np.array_equal(orc.a["adv_vbtiso"][:, :, 1:jmt - 1], g["adv_vbtiso"])
    tp = orc.transport()
    # bit-exact: same operations in the same order, no FMA contraction on either side
    assert np.array_equal(tp[:, :, 1:jmt - 1], g["t_taup1"][:, :, 1:jmt - 1])


@pytest.mark.parametrize("dims", [(14, 14, 6), (102, 102, 19)])
def test_transport_matches_compiled_reference(dims):
    if not refmodel.available("p2", *dims):
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    import refdriver
    oc = synthetic.make_ocean("p2", *dims)
    ro = refdriver.RefOcean(oc)
    orc = oracle_c.Oracle(oc, to=ro.v["to"], so=ro.v["so"], c=ro.v["c"])
    ro.isopyc(); orc.isopyc()
    jmt = dims[1]
    for name, sl in (("alphai", slice(None)), ("betai", slice(None)), ("ddzt", slice(None)),
                     ("ddxt", slice(1, jmt - 1)), ("ddyt", slice(0, jmt - 1)), ("Ai_ez", slice(1, jmt - 1)),
                     ("Ai_nz", slice(0, jmt - 1)), ("Ai_bx", slice(1, jmt - 1)), ("Ai_by", slice(1, jmt - 1)),
                     ("K11", slice(1, jmt - 1)), ("K22", slice(0, jmt - 1)), ("K33", slice(1, jmt - 1)),
                     ("adv_vetiso", slice(1, jmt - 1)), ("adv_vbtiso", slice(1, jmt - 1))):
        assert np.array_equal(orc.a[name][:, :, sl], ro.v[name.lower()]), name
    ro.add_k33(); orc.add_k33()
    ro.tracer()
    tp = orc.transport()
    assert np.array_equal(tp[:, :, 1:jmt - 1], ro.v["t"][:, :, 1:jmt - 1, :, 2])


def test_eos_fixture_matches_reference():
    if not refmodel.available("p2", 14, 14, 6):
        pytest.skip("oracle/_ref not built")
    ref = refmodel.RefLib("p2", 14, 14, 6)
    for km in (6, 19, 32):
        g = synthetic.make_grid(14, 14, km)
        zt = np.ascontiguousarray(g.zt)
        ro0, to, so = np.zeros(km), np.zeros(km), np.zeros(km)
        c = np.zeros((km, 9), order="F")
        w = [np.zeros(km) for _ in range(4)]
        ref.call("eqstate", zt, km, ro0, to, so, c, *w)
        to2, so2, c2 = synthetic.load_eos(km)
        assert np.array_equal(to, to2) and np.array_equal(so, so2) and np.array_equal(c, c2)


def test_oracle_threads_give_the_same_bits():
    """bench.py's courtesy N-core figure (cpu_baseline_ncore) shares the tracers of the transport, the rows of convct2 and the
    rows of the MOBI sources out over OpenMP threads: every value is computed by the same expressions, so the result must
    equal the one-thread result bit for bit (the parity tests all use one thread)."""
    import oracle_c
    import mobi_c
    from uvic29_amd import OPTION_SETS, mobi as pm, synthetic
    cfg = OPTION_SETS["c30"]
    oc = synthetic.make_ocean(cfg, 14, 14, 6)
    to, so, c = synthetic.load_eos(6)
    prm = pm.load_table("c30", 6)
    res = []
    try:
        for nth in (1, 4):
            oracle_c.lib().orc_set_threads(nth)
            orc = oracle_c.Oracle(oc, to=to, so=so, c=c, src=None)
            src = np.array(mobi_c.mobi_sources(oc, prm, oc.t_taum1, 2.0 * oc.params.dtts))
            orc.set_src(src)
            orc.isopyc(); orc.add_k33()
            res.append((src, np.array(orc.transport())))
    finally:
        oracle_c.lib().orc_set_threads(1)
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    assert np.abs(res[0][1]).max() > 0
