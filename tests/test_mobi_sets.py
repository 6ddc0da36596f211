"""MOBI option sets other than C: F (f18: C without the isotopes, with prognostic CaCO3) and the shipped
nt=37 set (s37: C + O_mobi_caco3 + O_mobi_silicon).  oracle/mobi_gen_oracle.c keeps the cpp options of mobi.F as run-time
flags; it is pinned here three ways:
  * with the flags of set C it reproduces oracle/mobi_oracle.c (itself pinned to the reference) bit for bit;
  * mobi_driver column by column against the COMPILED reference of f18 and s37, including trcmin collisions and
    suboxic columns (needs oracle/_ref);
  * the whole step (isopyc + sources + transport + convection) against the committed golden of the compiled reference.
Set E (e13) is not covered and cannot be: without O_mobi_alk the reference reads t(i,:,j,ialk,taum1) with ialk = 0
(tracer.F:491) -- out of bounds in the reference itself."""
import ctypes
from pathlib import Path

import numpy as np
import pytest

from uvic29_amd import synthetic, mobi as pm
import mobi_c
import mobi_gen_c
import oracle_c
import refmodel

GOLD = Path(__file__).resolve().parent / "golden"
SETS = ["f18", "s37"]


def test_generic_oracle_with_the_flags_of_set_c_equals_the_set_c_oracle():
    oc = synthetic.make_ocean("c30", 14, 14, 6)
    prm = pm.load_table("c30", 6)
    c2 = 2 * oc.params.dtts
    a = mobi_c.mobi_sources(oc, prm, oc.t_taum1, c2)
    b = mobi_gen_c.mobi_sources(oc, prm, oc.t_taum1, c2)
    assert np.abs(a).max() > 0 and np.array_equal(a, b)


@pytest.mark.parametrize("cfg", SETS)
def test_parameter_tables_and_column_order_match_reference_init(cfg):
    if not refmodel.available(cfg, 14, 14, 6):
        pytest.skip("oracle/_ref not built")
    import refdriver
    oc = synthetic.make_ocean(cfg, 14, 14, 6)
    ro = refdriver.RefOcean(oc)
    prm = pm.load_table(cfg, 6)
    for n, val in prm.items():
        if n == "imobi":
            for name, pos in val.items():
                assert oc.cfg.imobi(name) == pos == int(ro.v["imobi" + name][0]), name
            continue
        assert np.array_equal(np.atleast_1d(val), ro.v[n].reshape(-1)), n
    for name in oc.cfg.tracers:      # tracer and source-slot order of configs.py == tracer_init's
        assert int(ro.v["i" + name][0]) == oc.cfg.index(name), name
    for s, name in enumerate(oc.cfg.sources):
        assert int(ro.v["is" + name][0]) == s + 1, name


@pytest.mark.parametrize("cfg", SETS)
def test_oracle_step_matches_golden_of_the_compiled_reference(cfg):
    oc = synthetic.make_ocean(cfg, 14, 14, 6)
    to, so, c = synthetic.load_eos(6)
    prm = pm.load_table(cfg, 6)
    c2 = 2 * oc.params.dtts
    src = mobi_gen_c.mobi_sources(oc, prm, oc.t_taum1, c2)
    assert all(np.abs(src[..., s]).max() > 0 for s in range(oc.cfg.nsrc)), "a source slot is identically zero"
    orc = oracle_c.Oracle(oc, to=to, so=so, c=c, src=src)
    orc.isopyc(); orc.add_k33()
    tp = orc.transport()
    g = np.load(GOLD / f"step_{cfg}_14x14x6.npz")
    assert np.array_equal(tp[:, :, 1:13], g["t_taup1"][:, :, 1:13])


@pytest.mark.parametrize("cfg", SETS)
def test_mobi_driver_columns_match_compiled_reference(cfg):
    if not refmodel.available(cfg, 14, 14, 6):
        pytest.skip("oracle/_ref not built")
    import refdriver
    oc = synthetic.make_ocean(cfg, 14, 14, 6)
    cs, km = oc.cfg, 6
    ro = refdriver.RefOcean(oc)
    prm = pm.load_table(cfg, km)
    c2 = 2 * oc.params.dtts
    P = mobi_gen_c.make_params(cs, oc.grid, prm, c2)
    for n in ("nbio", "dtbio", "rdtts", "rnbio"):
        ro.ref.set(n, getattr(P, n))
    lib = oracle_c.lib()
    rng = np.random.default_rng(11)
    d = ctypes.c_double
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    checked = 0
    for trial in range(80):
        i, j = rng.integers(1, 13), rng.integers(1, 13)
        kmx = int(oc.topo.kmt[i, j])
        if kmx == 0:
            continue
        tn = np.zeros((km, cs.ntnpzd), order="F")
        for m, name in enumerate(cs.mobi):
            tn[:, m] = oc.t_taum1[i, :, j, cs.index(name) - 1]
        if trial % 3 == 0:      # collisions with trcmin: negative-prevention flags
            tn[rng.integers(0, km), rng.integers(0, cs.ntnpzd)] = 1e-13
        t_in = oc.t_taum1[i, :, j, 0].copy()
        o2 = oc.t_taum1[i, :, j, cs.index("o2") - 1] * 1000. * (0.02 if trial % 5 == 0 else 1.0)  # suboxic columns
        s_in = 1e3 * oc.t_taum1[i, :, j, 1] + 35.
        aou = 200.0 - o2
        dic = oc.t_taum1[i, :, j, cs.index("dic") - 1].copy()
        alk = oc.t_taum1[i, :, j, cs.index("alk") - 1].copy() * (0.93 if trial % 4 == 0 else 1.0)  # undersaturated: dissolution
        sgb = np.zeros(km); sgb[kmx - 1] = 1.0; sgb[max(kmx - 2, 0)] = 0.3
        tn1, tn2 = tn.copy(order="F"), tn.copy(order="F")
        s1, s2 = np.zeros((km, cs.nsrc), order="F"), np.zeros((km, cs.nsrc), order="F")
        ro.ref.call("mobi_driver", kmx, c2, 5e-4, 0.45, 90.0, tn1, t_in, o2, aou, s_in, dic, alk, 280.0, sgb, s1)
        lib.orc_mobig_driver(ctypes.byref(P), kmx, d(c2), d(5e-4), d(0.45), d(90.0), p(tn2), p(t_in), p(o2), p(aou),
                             p(s_in), p(dic), p(alk), d(280.0), p(sgb), p(s2))
        assert np.array_equal(s1, s2) and np.array_equal(tn1, tn2), trial
        checked += 1
    assert checked > 30


# ---- the device ---------------------------------------------------------------------------------------------
MOBI_RTOL = 1e-11     # device exp/log/pow/tanh differ from libm in the last bits (as for option set C, tests/test_mobi.py)


def _table(cfg, km):
    return pm.load_table(cfg, km)


@pytest.mark.gpu
@pytest.mark.parametrize("team", [True, False])
@pytest.mark.parametrize("cfg,dims", [("f18", (14, 14, 6)), ("s37", (14, 14, 6)), ("s37", (102, 102, 19)), ("f18", (102, 102, 19))])
def test_gpu_sources_of_the_other_option_sets_vs_oracle(cfg, dims, team):
    """uvic_gpu_set_mobi_opt + uvic_gpu_mobi against the run-time-flag oracle: the four-wave-team kernels of
    csrc/kernels_mobi_gt.hpp (the default) and the one-thread-per-column kernel of kernels_mobi_gen.hpp (the cross-check)."""
    from uvic29_amd.tracer import TracerModel
    oc = synthetic.make_ocean(cfg, *dims)
    prm = _table(cfg, dims[2])
    to, so, c = synthetic.load_eos(dims[2])
    want = mobi_gen_c.mobi_sources(oc, prm, oc.t_taum1, 2 * oc.params.dtts)
    m = TracerModel(*dims, oc.cfg.nt, oc.cfg.nsrc, oc.cfg.ntnpzd)
    m.load_ocean(oc, to, so, c)
    m.set_option("mobi_team", 1 if team else 0)
    m.set_mobi(oc)
    m.mobi()
    got = m.download("src")
    worst = 0.0
    for s, name in enumerate(oc.cfg.sources):
        scale = np.abs(want[..., s]).max()
        err = np.abs(got[..., s] - want[..., s]).max()
        worst = max(worst, err / scale)
        assert scale > 0 and err <= MOBI_RTOL * scale, (name, err, scale)
    print("worst relative source error", cfg, dims, worst)
    m.close()


@pytest.mark.gpu
def test_gpu_general_kernel_with_the_flags_of_set_c_vs_the_set_c_oracle():
    """The general kernel forced onto option set C (set_option "mobi_generic") against the oracle that is pinned hardest."""
    from uvic29_amd.tracer import TracerModel
    oc = synthetic.make_ocean("c30", 14, 14, 6)
    prm = pm.load_table("c30", 6)
    to, so, c = synthetic.load_eos(6)
    want = mobi_c.mobi_sources(oc, prm, oc.t_taum1, 2 * oc.params.dtts)
    m = TracerModel(14, 14, 6, oc.cfg.nt, oc.cfg.nsrc, oc.cfg.ntnpzd)
    m.load_ocean(oc, to, so, c)
    m.set_mobi(oc, generic=True)
    m.mobi()
    got = m.download("src")
    for s, name in enumerate(oc.cfg.sources):
        scale = np.abs(want[..., s]).max()
        assert np.abs(got[..., s] - want[..., s]).max() <= MOBI_RTOL * scale, name
    m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", SETS)
def test_gpu_full_step_of_the_other_option_sets_vs_golden(cfg):
    """One complete `tracer` step (MOBI + transport + convection, bit-exact transport) against the golden output of the
    compiled reference; T and S have no biological source: bit-exact."""
    from uvic29_amd.tracer import TracerModel
    oc = synthetic.make_ocean(cfg, 14, 14, 6)
    to, so, c = synthetic.load_eos(6)
    g = np.load(GOLD / f"step_{cfg}_14x14x6.npz")
    m = TracerModel(14, 14, 6, oc.cfg.nt, oc.cfg.nsrc, oc.cfg.ntnpzd)
    m.set_exact(True)
    m.load_ocean(oc, to, so, c)
    m.set_mobi(oc)
    m.isopyc(); m.tracer()
    got = m.download("t_taup1")
    for n, name in enumerate(oc.cfg.tracers):
        a, b = got[:, :, 1:13, n], g["t_taup1"][:, :, 1:13, n]
        assert np.abs(a - b).max() <= 1e-11 * np.abs(b).max(), (name, np.abs(a - b).max(), np.abs(b).max())
    assert np.array_equal(got[:, :, 1:13, :2], g["t_taup1"][:, :, 1:13, :2])
    m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", SETS)
def test_gpu_time_loop_of_the_other_option_sets_lookahead_equals_inline(cfg):
    """Ten steps of the production loop (sources one step ahead on the side streams, a forward step inside) against the
    same steps with everything in line: the look-ahead machinery does not care which MOBI kernel runs."""
    from uvic29_amd.tracer import TracerModel, TimeLoop
    oc = synthetic.make_ocean(cfg, 14, 14, 6)
    to, so, c = synthetic.load_eos(6)
    out = {}
    for how in ("ahead", "inline"):
        m = TracerModel(14, 14, 6, oc.cfg.nt, oc.cfg.nsrc, oc.cfg.ntnpzd)
        m.load_ocean(oc, to, so, c)
        m.set_mobi(oc)
        loop = TimeLoop(m, oc.params.dtts, 4, segment=0 if how == "ahead" else 1)
        for _ in range(10):
            loop.step()
        m.sync()
        out[how] = m.download("t_tau")
        m.close()
    assert np.isfinite(out["ahead"]).all() and np.array_equal(out["ahead"], out["inline"])
