"""The reference's second boundary: `tracer` compiled with -DO_TMM is a column-batch source operator
(imt = batch size, jmt = 1; u09/mom/tracer.F:109-124).  CPU: the C restatement's sources for the same columns equal the
reference built that way (oracle/_ref build "tmm30"), bit for bit.  GPU: uvic_gpu_tmm_* == the restatement."""
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
for p in (ROOT, ROOT / "oracle"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))


def _batch(cfg="c30", imt=14, jmt=14, km=6, ncols=64):
    from uvic29_amd import synthetic
    oc = synthetic.make_ocean(cfg, imt, jmt, km)
    cols = [(i, j) for j in range(2, jmt) for i in range(2, imt)][:ncols]     # ocean and land columns alike
    return oc, cols


def _oracle_sources(oc, cols, c2dtts):
    import mobi_c
    from uvic29_amd import mobi as pm
    full = mobi_c.mobi_sources(oc, pm.load_table(oc.cfg.name, oc.grid.km), oc.t_taum1, c2dtts)
    return np.stack([full[i - 1, :, j - 1, :] for (i, j) in cols])


def test_oracle_equals_the_reference_built_with_O_TMM():
    import refdriver
    import refmodel
    oc, cols = _batch()
    if not refmodel.available("tmm30", len(cols), 1, oc.grid.km):
        pytest.skip("oracle/_ref build tmm30 not present")
    c2dtts = 2.0 * oc.params.dtts
    got = refdriver.tmm_reference_sources(oc, cols, c2dtts)
    want = _oracle_sources(oc, cols, c2dtts)
    assert np.abs(want).max() > 1e-8 and np.array_equal(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,tol", [("c30", 1e-11), ("s37", 1e-11)])
def test_gpu_column_batch_operator(cfg, tol):
    from uvic29_amd import mobi as pm
    from uvic29_amd.tmm import TmmOperator
    oc, cols = _batch(cfg, ncols=100)
    g, f, topo = oc.grid, oc.forcing, oc.topo
    ii = np.array([c[0] - 1 for c in cols]); jj = np.array([c[1] - 1 for c in cols])
    op = TmmOperator(oc.cfg, g, len(cols))
    op.set_columns(topo.kmt[ii, jj], g.tlat[ii, jj], topo.sg_bathy[ii, jj, :], f.fe_atmdep[ii, jj, :], f.fe_hydr[ii, jj, :],
                   f.dnswr[ii, jj], f.aice[ii, jj], f.hice[ii, jj], f.hsno[ii, jj], f.relyr, f.co2ccn)
    for c2dtts in (2.0 * oc.params.dtts, oc.params.dtts):
        got = op.sources(oc.t_taum1[ii, :, jj, :], c2dtts)
        if cfg == "c30":
            want = _oracle_sources(oc, cols, c2dtts)
        else:
            import mobi_gen_c
            full = mobi_gen_c.mobi_sources(oc, pm.load_table(cfg, g.km), oc.t_taum1, c2dtts)
            want = np.stack([full[i - 1, :, j - 1, :] for (i, j) in cols])
        scale = np.abs(want).max(axis=(0, 1), keepdims=True) + 1e-300
        assert (np.abs(got - want) / scale).max() <= tol
    # a second batch through the same handle: the other half of the grid, new forcing
    cols2 = [(i, j) for j in range(2, 14) for i in range(2, 14)][44:144]
    i2 = np.array([c[0] - 1 for c in cols2]); j2 = np.array([c[1] - 1 for c in cols2])
    op.set_columns(topo.kmt[i2, j2], g.tlat[i2, j2], topo.sg_bathy[i2, j2, :], f.fe_atmdep[i2, j2, :], f.fe_hydr[i2, j2, :],
                   f.dnswr[i2, j2], f.aice[i2, j2], f.hice[i2, j2], f.hsno[i2, j2], f.relyr, f.co2ccn)
    got = op.sources(oc.t_taum1[i2, :, j2, :], 2.0 * oc.params.dtts)
    if cfg == "c30":
        want = _oracle_sources(oc, cols2, 2.0 * oc.params.dtts)
        scale = np.abs(want).max(axis=(0, 1), keepdims=True) + 1e-300
        assert (np.abs(got - want) / scale).max() <= tol
    op.close()
