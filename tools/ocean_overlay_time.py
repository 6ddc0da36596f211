#!/usr/bin/env python3
"""Wall time per time step of the two Fortran overlays together -- `tracer` then `clinic`, as source/mom/mom.F:389-395
calls them -- on the 102x102x19 grid (oracle/_ref shim "m2": T and S only; or "t30": option set C built as run/mk.in builds it, with tsiperts on every step
and ocean segments of four steps), PCIe included, with the velocities shipped on
every call (UVIC_RESIDENT=1) or resident on the device (UVIC_RESIDENT=2: psi and the wind stress up, zu down).  The calls
of `isopyc` and `vmixc` (mom.F:340-347, with the tidal mixing of run/mk.in) are timed beside them: the reference's own
routines on one host core, or -- UVIC_RESIDENT=3, mixing_gpu.F -- two calls that return at once because the device forms
the tensor and diff_cbt itself.  The other host routines of the loop (loadmw's add_ext_mode, state, adv_vel, setvbc) are
run but not timed.
usage: UVIC_RESIDENT=1|2|3 python tools/ocean_overlay_time.py [nsteps [m2|t30]] [--tavg] [--notsi] [--json]"""
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
for p in (ROOT, ROOT / "oracle", ROOT / "tests"):
    sys.path.insert(0, str(p))
import numpy as np  # noqa: E402
from uvic29_amd import synthetic  # noqa: E402
import refdriver  # noqa: E402
from test_clinic import _psi  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
n = int(args[0]) if args else 24
cfg = args[1] if len(args) > 1 else "m2"       # "t30": option set C built as run/mk.in builds it, tsiperts on every step
as_json, seg = "--json" in sys.argv, 4
notsi = "--notsi" in sys.argv                  # t30 without the time-step monitor (tsiint > one step): the call returns with T and S
tavg = "--tavg" in sys.argv                    # every step a time-average step as well (timavgperts: one year in ten of the shipped run)
dims = (102, 102, 19)
oc = synthetic.make_ocean(cfg, *dims)
mom = synthetic.make_momentum(oc.grid, oc.topo, oc.u, anisotropic=True)
g = oc.grid
R = refdriver.RefOcean(oc, shim=True)
R.set_momentum(mom)
R.set_filter(synthetic.make_filter(g, g.km)); R.set_filter_u(synthetic.make_filter_u(g, g.km))
S, v = R.ref.set, R.v
np_ = v["sbc"].shape[2]
S("ihflx", np_ - 3); S("isflx", np_ - 2)
v["sbc"][:, :, np_ - 4] = oc.stf[:, :, 0]; v["sbc"][:, :, np_ - 3] = oc.stf[:, :, 1]
R.set_step_kind(False)
v["u"][..., 2] = 0.0
if cfg == "t30":
    S("nmix", 0); S("ntspos", seg); S("prelyr", float(v["relyr"][0])); S("tsiperts", 0 if notsi else 1)
    S("timavgperts", 1 if tavg else 0)
tid = synthetic.make_tidal(g, oc.topo, oc.params.kappa_h)
for nm in ("edrm2", "edrs2", "edrk1", "edro1"):
    v[nm][...] = getattr(tid, nm)
for nm in ("zetar", "ogamma", "gravrho0r", "kappa_h"):
    S(nm, getattr(tid, nm))
v["diff_cbt"][...] = oc.diff_cbt_bg[:, :, 1:g.jmt - 1]
tt, tc, tm, ta, tst = [], [], [], [], []
for it in range(1, n + 1):
    S("itt", it)
    if cfg == "t30":
        S("osegs", 1 if (it - 1) % seg == 0 else 0); S("osege", 1 if it % seg == 0 else 0)
        for nm in ("tbar", "travar", "dtabs", "ektot"):      # diagi zeroes them at the start of every step
            v[nm][...] = 0.0
    R.add_ext_mode(_psi(g, it), "tau")
    if it == 1:
        R.add_ext_mode(_psi(g, 0), "tau-1")
    ts0 = time.perf_counter()
    R.state()
    tst.append(time.perf_counter() - ts0)
    ta0 = time.perf_counter()
    R.ref.call("adv_vel", 0, 1, g.jmt, 2, g.imt - 1)
    ta.append(time.perf_counter() - ta0)
    tm0 = time.perf_counter()
    R.isopyc()
    R.ref.call("vmixc", 0, 1, g.jmt, 2, g.imt - 1)
    tm.append(time.perf_counter() - tm0)
    R.setvbc()
    t0 = time.perf_counter()
    R.ref.call("tracer", 0, 2, g.jmt - 1, 2, g.imt - 1)
    t1 = time.perf_counter()
    R.ref.call("clinic", 0, 2, g.jmt - 1, 2, g.imt - 1)
    t2 = time.perf_counter()
    tt.append(t1 - t0); tc.append(t2 - t1)
    R.rotate()
    u = v["u"]; u[..., 0] = u[..., 1]; u[..., 1] = u[..., 2]
med = lambda x: sorted(x[4:])[len(x[4:]) // 2] * 1e3
if as_json:
    import json
    print(json.dumps({"tracer_call_ms": med(tt), "clinic_call_ms": med(tc), "isopyc_vmixc_calls_ms": med(tm), "adv_vel_call_ms": med(ta), "state_call_ms": med(tst), "steps": n, "cfg": cfg, "time_average_steps": tavg,
                      "resident": os.environ.get("UVIC_RESIDENT", "")}))
    sys.exit(0)
print(f"UVIC_RESIDENT={os.environ.get('UVIC_RESIDENT', '')}: tracer call {med(tt):.3f} ms, clinic call {med(tc):.3f} ms, isopyc + vmixc calls {med(tm):.3f} ms, adv_vel call {med(ta):.3f} ms, state call {med(tst):.3f} ms (medians over {n - 4} steps, PCIe included)")
