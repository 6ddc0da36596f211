"""Polar Fourier filter of the tracers (source/common/filt.F, filtr.F, findex.F).

CPU: C restatement == compiled reference (findex strips and filtered field, bit for bit), also on a
topography carved so that the filtered rows hold partial strips, strips that wrap around the cyclic
boundary and full circles of different depths; library set-up + workgroup routine under host
emulation == C restatement.  GPU: `tracer` with the filter on == oracle transport + filter."""
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
for p in (ROOT, ROOT / "oracle", ROOT / "tests" / "hostemu"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

GRIDS = [("p2", 14, 14, 6), ("p2", 102, 102, 19)]


def carve(oc, flt):
    """Open the filtered rows and put land blocks of varying depth into them."""
    g, kmt = oc.grid, oc.topo.kmt
    imt, jmt, km = g.imt, g.jmt, g.km
    rows = [j for j in range(flt.jfrst, jmt) if j <= flt.jft1 or j >= flt.jft2]
    for n, j in enumerate(rows):
        i = np.arange(2, imt)
        depth = km - (n % 3) - ((i + 2 * n) % 5 == 0)                       # ragged bottom
        depth = np.where(((i + 3 * n) % max(7, imt // 6)) < 2, (n % 2) * (km // 2), depth)   # land or shallow blocks
        if n % 4 == 1:
            depth[: max(2, imt // 10)] = km                                  # ocean across the cyclic boundary ...
            depth[-max(2, imt // 12):] = km
            depth[imt // 3: imt // 3 + 3] = 0                                # ... but not a full circle
        if n % 4 == 3:
            depth[:] = km - 1                                                # a full circle
        kmt[1:imt - 1, j - 1] = np.maximum(depth, 0)
    kmt[0] = kmt[imt - 2]
    kmt[imt - 1] = kmt[1]


def _setup(cfg, imt, jmt, km, carved):
    from uvic29_amd import synthetic
    oc = synthetic.make_ocean(cfg, imt, jmt, km)
    flt = synthetic.make_filter(oc.grid, km)
    if carved:
        carve(oc, flt)
    rng = np.random.default_rng(5)
    field = np.asfortranarray(oc.t_tau * (1.0 + 0.05 * rng.standard_normal(oc.t_tau.shape)))
    field[0] = field[-2]          # cyclic images consistent, as every field of the model is
    field[-1] = field[1]
    return oc, flt, field


@pytest.mark.parametrize("carved", [False, True])
@pytest.mark.parametrize("cfg,imt,jmt,km", GRIDS)
def test_oracle_equals_reference(cfg, imt, jmt, km, carved):
    import oracle_c
    import refmodel
    if not refmodel.available(cfg, imt, jmt, km):
        pytest.skip("oracle/_ref not built for this grid")
    import refdriver
    oc, flt, field = _setup(cfg, imt, jmt, km, carved)
    ro = refdriver.RefOcean(oc)
    if "findex_" not in dir(ro.ref.lib) and not hasattr(ro.ref.lib, "findex_"):
        pytest.skip("oracle/_ref predates findex")
    ristf, rietf = ro.set_filter(flt)
    istf, ietf = oracle_c.findex(oc.topo.kmt, flt)
    assert np.array_equal(istf, ristf) and np.array_equal(ietf, rietf)
    if carved:
        im = np.where(istf > 0, ietf - istf + 1, 0)
        assert (im == imt - 2).any() and ((im > 0) & (im < imt - 2)).any() and (ietf >= imt).any()   # circles, partial, wrapping
    ro.v["t"][..., 2] = field
    want = np.array(ro.filt(), order="F")
    got = oracle_c.filt(field.copy(order="F"), oc.grid, oc.topo, flt, istf, ietf)
    assert np.array_equal(got[:, :, 1:-1], want[:, :, 1:-1])
    assert (want[:, :, 1:-1] != field[:, :, 1:-1]).any()


@pytest.mark.parametrize("carved", [False, True])
@pytest.mark.parametrize("cfg,imt,jmt,km", GRIDS)
def test_hostemu_equals_oracle(cfg, imt, jmt, km, carved):
    import emu
    import oracle_c
    from uvic29_amd import synthetic
    oc, flt, field = _setup(cfg, imt, jmt, km, carved)
    istf, ietf = oracle_c.findex(oc.topo.kmt, flt)
    want = oracle_c.setbcx(oracle_c.filt(field.copy(order="F"), oc.grid, oc.topo, flt, istf, ietf))   # tracer.F:1252 after filt
    em = emu.EmuOcean(oc, *synthetic.load_eos(km))
    em.a["t_taup1"] = field.copy(order="F")
    em.rebind()
    n = em.filt(flt)
    assert n == int((istf > 0).sum())
    assert np.array_equal(em.a["t_taup1"][:, :, 1:-1], want[:, :, 1:-1])


def test_reference_tracer_with_filter_equals_oracle_transport_then_filter():
    import oracle_c
    import refmodel
    if not refmodel.available("p2", 102, 102, 19):
        pytest.skip("oracle/_ref not built")
    import refdriver
    from uvic29_amd import synthetic
    oc, flt, _ = _setup("p2", 102, 102, 19, False)
    to, so, c = synthetic.load_eos(19)
    ro = refdriver.RefOcean(oc)
    if not hasattr(ro.ref.lib, "findex_"):
        pytest.skip("oracle/_ref predates findex")
    istf, ietf = ro.set_filter(flt)
    want = np.array(ro.step(), order="F")
    orc = oracle_c.Oracle(oc, to=to, so=so, c=c)
    orc.isopyc(); orc.add_k33()
    got = oracle_c.setbcx(oracle_c.filt(np.array(orc.transport(), order="F"), oc.grid, oc.topo, flt, istf, ietf))
    assert np.array_equal(got[:, :, 1:-1], want[:, :, 1:-1])


@pytest.mark.gpu
@pytest.mark.parametrize("exact", [True, False])
@pytest.mark.parametrize("cfg,imt,jmt,km", GRIDS + [("p2", 23, 17, 6), ("p2", 38, 101, 19)])     # (not square, odd)
def test_gpu_tracer_with_filter(cfg, imt, jmt, km, exact):
    import oracle_c
    from uvic29_amd import synthetic
    from uvic29_amd.tracer import TracerModel
    oc, flt, _ = _setup(cfg, imt, jmt, km, False)
    to, so, c = synthetic.load_eos(km)
    orc = oracle_c.Oracle(oc, to=to, so=so, c=c)
    orc.isopyc(); orc.add_k33()
    plain = np.array(orc.transport(), order="F")
    istf, ietf = oracle_c.findex(oc.topo.kmt, flt)
    want = oracle_c.setbcx(oracle_c.filt(plain.copy(order="F"), oc.grid, oc.topo, flt, istf, ietf))
    assert (want != plain).any()
    m = TracerModel(imt, jmt, km, oc.cfg.nt, oc.cfg.nsrc, oc.cfg.ntnpzd, device=0)
    m.load_ocean(oc, to, so, c)
    m.set_exact(exact)
    m.set_filter(oc, flt)
    m.isopyc(); m.tracer()
    got = m.download("t_taup1")
    if exact:
        assert np.array_equal(got[:, :, 1:-1], want[:, :, 1:-1])
    else:
        scale = np.abs(want).max(axis=(0, 1, 2), keepdims=True)
        assert (np.abs(got - want)[:, :, 1:-1] / scale).max() <= 1e-13
    m.set_filter(oc, None)                      # off again
    m.load_ocean(oc, to, so, c)
    m.isopyc(); m.tracer()
    off = m.download("t_taup1")
    if exact:
        assert np.array_equal(off[:, :, 1:-1], plain[:, :, 1:-1])
    m.close()


@pytest.mark.gpu
def test_gpu_filter_on_carved_topography_equals_oracle():
    """The filter alone (convection finds nothing to mix in a uniform field ... so feed it through convect):
    upload a field as t(tau+1), run the post-transport part, compare with the oracle filter."""
    import oracle_c
    from uvic29_amd import synthetic
    from uvic29_amd.tracer import TracerModel
    oc, flt, field = _setup("p2", 102, 102, 19, True)
    to, so, c = synthetic.load_eos(19)
    # a stably stratified copy so that convection leaves it alone: filter result must equal the oracle's
    strat = np.asfortranarray(field.copy())
    strat[..., 0] = (30.0 - np.arange(19) * 1.2)[None, :, None] + 0.01 * field[..., 0]
    strat[..., 1] = 0.0
    istf, ietf = oracle_c.findex(oc.topo.kmt, flt)
    want = oracle_c.setbcx(oracle_c.filt(strat.copy(order="F"), oc.grid, oc.topo, flt, istf, ietf))
    m = TracerModel(102, 102, 19, 2, 0, 0, device=0)
    m.load_ocean(oc, to, so, c)
    m.set_filter(oc, flt)
    m.upload("t_taup1", strat)
    m.convect()
    got = m.download("t_taup1")
    rows = [j for j in range(flt.jfrst, 102) if j <= flt.jft1 or j >= flt.jft2]
    J = np.array(rows) - 1
    assert np.array_equal(got[:, :, J], want[:, :, J])
    m.close()
