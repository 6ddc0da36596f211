#!/usr/bin/env python3
"""Golden 100-step run of the COMPILED REFERENCE (oracle/_ref) on the full-size synthetic ocean.

c30 (nt = 30, MOBI), 102x102x19, 100 ocean steps with a forward ("mixing") step every nmix-th, constant forcing -- the
drift workload of SURVEY.md 8(d).  Stored (tests/golden/run_c30_102x102x19_n100.npz, data only):
  tbar, travar : per tracer, the volume integrals of t(tau) and t(tau)**2 the reference's time-step monitor forms
                 (updates/09/source/mom/tracer.F:1516-1537: darea = dzt*dxt*cst*dyt*tmask), summed over k and rows
  tmax         : per tracer max|t| over the ocean
  cols_i, cols_j, cols : 90 sample columns (every 10th column of every 10th ocean row), all levels, all tracers
Run in the build container (needs /root/reference for mobi_init's control.in):  python tests/golden/make_golden_run100.py
"""
import sys
import time
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
for p in (ROOT, ROOT / "oracle"):
    sys.path.insert(0, str(p))

from uvic29_amd import synthetic  # noqa: E402
import refdriver  # noqa: E402

NSTEPS = 100


def sample_columns(g):
    ii = np.arange(6, g.imt - 1, 10)
    jj = np.arange(12, g.jmt - 10, 10)
    I, J = np.meshgrid(ii, jj, indexing="ij")
    return I.ravel(), J.ravel()      # 1-based


def integrals(oc, t):
    """tbar, travar of tracer.F:1516-1537 summed over levels and rows; columns is..ie = 2..imt-1, rows 2..jmt-1"""
    g, topo = oc.grid, oc.topo
    darea = (g.dzt[None, :, None] * g.dxt[:, None, None] * (g.cst * g.dyt)[None, None, :]) * topo.tmask
    sl = (slice(1, g.imt - 1), slice(None), slice(1, g.jmt - 1))
    tb = np.array([(t[..., n][sl] * darea[sl]).sum() for n in range(t.shape[3])])
    tv = np.array([((t[..., n][sl] ** 2) * darea[sl]).sum() for n in range(t.shape[3])])
    return tb, tv


def reference_run(oc, nsteps=NSTEPS, progress=False):
    """The loop of `mom` on the reference's own routines (as tests/golden/make_golden.py:run_fixture)."""
    ro = refdriver.RefOcean(oc)
    dtts, nmix = oc.params.dtts, oc.params.nmix
    t0 = time.time()
    for it in range(1, nsteps + 1):
        mixing = (it % nmix) == 0
        if mixing:                      # forward step: both slots hold tau (loadmw.F:107-111)
            ro.v["t"][..., 0] = ro.v["t"][..., 1]
        ro.step(c2dtts=dtts if mixing else 2.0 * dtts)
        ro.rotate()
        if progress and it % 10 == 0:
            print(f"  reference step {it} ({time.time() - t0:.0f} s)", flush=True)
    return np.array(ro.v["t"][..., 1], order="F")


def main():
    oc = synthetic.make_ocean("c30")
    t = reference_run(oc, progress=True)
    tb, tv = integrals(oc, t)
    ci, cj = sample_columns(oc.grid)
    cols = np.stack([t[i - 1, :, j - 1, :] for i, j in zip(ci, cj)])      # (ncol, km, nt)
    tmax = np.abs(t[:, :, 1:oc.grid.jmt - 1]).max(axis=(0, 1, 2))
    out = HERE / "run_c30_102x102x19_n100.npz"
    np.savez_compressed(out, tbar=tb, travar=tv, tmax=tmax, cols_i=ci, cols_j=cj, cols=cols, nsteps=NSTEPS)
    print("wrote", out, out.stat().st_size, "bytes")


if __name__ == "__main__":
    main()
