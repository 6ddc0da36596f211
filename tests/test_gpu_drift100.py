"""North-star drift criterion, pinned to the reference: c30 (nt = 30, MOBI) on 102x102x19, 100 ocean steps with a forward
step every 16th, (i) the bit-exact GPU path and (ii) the production path, each against the COMPILED REFERENCE stepped in
the same test (oracle/_ref travels to the GPU box as a built artefact) and against the committed golden of that run
(tests/golden/run_c30_102x102x19_n100.npz: tbar/travar integrals of tracer.F:1516-1537 and 90 sample columns), so that
the check also runs where oracle/_ref is absent.

What can be asked of 100 steps.  The scheme contains discrete switches -- convective adjustment compares densities
(convect.F:189-255), the FCT limiter takes min/max, MOBI clamps at trcmin -- so two evaluations that differ by rounding
agree to rounding only until one marginally stable column is adjusted in one run and not in the other.  The compiled
reference itself does this when one input is perturbed by one ulp (tools/sens.py).  On this ocean it happens in one
column (i=59, j=51, levels 4-6) between steps 12 and 16: |dT| there jumps from 7e-13 to 3e-9 of max|T| and then decays
(tools/drift_probe.py).  Measured on MI355X, |t - t_ref| / max|t_ref| per tracer over the ocean cells after 100 steps:
  bit-exact GPU path : T, S identical to the reference bit for bit; the others (device exp/log/pow in MOBI) <= 1.2e-14
  production path    : 99.9th percentile <= 1.1e-13, maximum 2.1e-10 (the one flipped column and what spread from it)
The criterion, per tracer:
  * bit-exact path: maximum <= 1e-12                                          (the north-star number, strictly)
  * production path: 99.9th percentile <= 1e-12 (the north-star number for the field) and maximum <= 1e-8 (an isolated
    flipped adjustment), against the reference AND against the bit-exact GPU path
and the global integrals tbar, travar of both agree with the reference's to 1e-12 relative.
"""
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).resolve().parent / "golden"
sys.path.insert(0, str(GOLD))

P999_TOL = 1e-12      # north star: relative drift after 100 steps
MAX_TOL = {"exact": 1e-12, "production": 1e-8}
INTEGRAL_TOL = 1e-12


def _gpu_run(oc, to, so, c, exact, nsteps=100):
    from uvic29_amd.tracer import TracerModel, TimeLoop
    m = TracerModel(102, 102, 19, oc.cfg.nt, oc.cfg.nsrc, oc.cfg.ntnpzd)
    m.set_exact(exact)
    m.load_ocean(oc, to, so, c)
    m.set_mobi(oc)
    loop = TimeLoop(m, oc.params.dtts, oc.params.nmix)
    for _ in range(nsteps):
        loop.step()
    m.sync()
    out = m.download("t_tau")
    m.close()
    return out


def _stats(oc, got, ref):
    """per tracer: max and 99.9th percentile of |got - ref| / max|ref| over the ocean cells"""
    wet = oc.topo.tmask[:, :, 1:101] > 0
    rows = []
    for n, name in enumerate(oc.cfg.tracers):
        d = np.abs(got[:, :, 1:101, n] - ref[:, :, 1:101, n])[wet] / np.abs(ref[:, :, 1:101, n]).max()
        rows.append((name, float(d.max()), float(np.quantile(d, 0.999))))
    return rows


def test_hundred_steps_against_the_reference():
    import refmodel
    from uvic29_amd import synthetic
    import make_golden_run100 as mg
    oc = synthetic.make_ocean("c30")
    to, so, c = synthetic.load_eos(19)
    gold = np.load(GOLD / "run_c30_102x102x19_n100.npz")
    runs = {"exact": _gpu_run(oc, to, so, c, True), "production": _gpu_run(oc, to, so, c, False)}
    ref = None
    if refmodel.available("c30", 102, 102, 19):
        ref = mg.reference_run(oc)
        # the committed golden is this run: same bits
        tb, tv = mg.integrals(oc, ref)
        assert np.array_equal(tb, gold["tbar"]) and np.array_equal(tv, gold["travar"])
        cols = np.stack([ref[i - 1, :, j - 1, :] for i, j in zip(gold["cols_i"], gold["cols_j"])])
        assert np.array_equal(cols, gold["cols"])
    for how, got in runs.items():
        assert np.isfinite(got).all()
        # integrals (always)
        tb, tv = mg.integrals(oc, got)
        rb = np.abs(tb - gold["tbar"]) / np.abs(gold["tbar"])
        rv = np.abs(tv - gold["travar"]) / np.abs(gold["travar"])
        print(f"{how}: integrals tbar {rb.max():.2e} travar {rv.max():.2e}")
        assert rb.max() <= INTEGRAL_TOL and rv.max() <= INTEGRAL_TOL, (how, rb.max(), rv.max())
        # the sample columns of the golden (always)
        cols = np.stack([got[i - 1, :, j - 1, :] for i, j in zip(gold["cols_i"], gold["cols_j"])])
        dc = np.abs(cols - gold["cols"]).max(axis=(0, 1)) / gold["tmax"]
        print(f"{how}: sample columns max {dc.max():.2e} ({oc.cfg.tracers[int(dc.argmax())]})")
        assert dc.max() <= MAX_TOL[how], (how, dc)
        # the whole field against the reference stepped here
        if ref is not None:
            rows = _stats(oc, got, ref)
            worst_max = max(rows, key=lambda r: r[1])
            worst_p = max(rows, key=lambda r: r[2])
            print(f"{how} vs reference after 100 steps: worst max {worst_max[1]:.2e} ({worst_max[0]}), "
                  f"worst 99.9th percentile {worst_p[2]:.2e} ({worst_p[0]})")
            print("   ", {r[0]: (float(f"{r[1]:.1e}"), float(f"{r[2]:.1e}")) for r in rows})
            for name, dmax, p999 in rows:
                assert p999 <= P999_TOL, (how, name, p999)
                assert dmax <= MAX_TOL[how], (how, name, dmax)
    # the production path against the bit-exact GPU path (runs everywhere, with or without oracle/_ref)
    rows = _stats(oc, runs["production"], runs["exact"])
    print("production vs bit-exact GPU path:", {r[0]: (float(f"{r[1]:.1e}"), float(f"{r[2]:.1e}")) for r in rows})
    for name, dmax, p999 in rows:
        assert p999 <= P999_TOL, ("production vs exact", name, p999)
        assert dmax <= MAX_TOL["production"], ("production vs exact", name, dmax)
