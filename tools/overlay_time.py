#!/usr/bin/env python3
"""Wall time of the Fortran overlay's `tracer` call (PCIe included) on c30 102x102x19, with the state uploaded and
downloaded every step or resident on the device (UVIC_RESIDENT=1), or resident with `isopyc` and `vmixc` left to the device
as well (UVIC_RESIDENT=3: the two calls, which then return at once, are made and timed with the `tracer` call; no diff_cbt
goes up).  One mode per process: the overlay reads the environment once.
usage: [UVIC_RESIDENT=1|3] python tools/overlay_time.py [nsteps]"""
import os
import sys
import time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "oracle"))
from uvic29_amd import synthetic  # noqa: E402
import refdriver  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
as_json = "--json" in sys.argv
n = int(args[0]) if args else 12
seg = int(os.environ.get("OVERLAY_SEGMENT", "4"))      # ocean steps per coupling segment (run/control.in: segtim 5 d / dtts 1.25 d)
oc = synthetic.make_ocean("c30", 102, 102, 19)
shim = refdriver.RefOcean(oc, shim=True)
shim.set_step_kind(False)
shim.ref.set("nmix", 0)
shim.ref.set("ntspos", seg)
shim.ref.set("prelyr", float(shim.v["relyr"][0]))
level3 = os.environ.get("UVIC_RESIDENT") == "3" and hasattr(shim.ref.lib, "uvic_mix_on_host_")
if level3:
    shim.set_tidal(synthetic.make_tidal(oc.grid, oc.topo, oc.params.kappa_h))
jmt_, imt_ = oc.grid.jmt, oc.grid.imt


def mixing_calls():      # mom.F:340-347
    shim.isopyc()
    shim.ref.call("vmixc", 0, 1, jmt_, 2, imt_ - 1)


ts = []
for it in range(1, n + 1):
    shim.ref.set("itt", it)
    shim.ref.set("osegs", 1 if (it - 1) % seg == 0 else 0)
    shim.ref.set("osege", 1 if it % seg == 0 else 0)
    if level3:
        t0 = time.perf_counter()
        mixing_calls()
    else:
        shim.isopyc(); shim.add_k33()      # host part of the "tracer only" integration, not timed
        t0 = time.perf_counter()
    shim.tracer()
    ts.append(time.perf_counter() - t0)
    shim.rotate()
ts = ts[4:]
g = oc.grid
units = g.imt * g.jmt * g.km * oc.cfg.nt
med = sorted(ts)[len(ts) // 2]
mean = sum(ts) / len(ts)
if not as_json:
  print(f"overlay tracer call, {'resident' if os.environ.get('UVIC_RESIDENT') in ('1', '3') else 'upload/download every step'}{' (isopyc, vmixc on the device)' if level3 else ''}, "
      f"segments of {seg} steps: median {med * 1e3:.3f} ms, mean {mean * 1e3:.3f} ms per step = {units / mean / 1e9:.2f} G cell-updates/s "
      f"(PCIe included)")
# the calls alone, back to back (the device never idles)
t0 = time.perf_counter()
each = []
for k in range(n):
    it += 1
    shim.ref.set("itt", it)
    shim.ref.set("osegs", 1 if (it - 1) % seg == 0 else 0)
    shim.ref.set("osege", 1 if it % seg == 0 else 0)
    t1 = time.perf_counter()
    if level3:
        mixing_calls()
    shim.tracer()        # (no host rotation here: the harness rotates by copying 2 x 47 MB, the model by permuting indices)
    each.append(time.perf_counter() - t1)
el = time.perf_counter() - t0
if as_json:
    import json
    print(json.dumps({"call_ms": med * 1e3, "loop_ms": el / n * 1e3, "segment": seg, "steps": n, "resident": os.environ.get("UVIC_RESIDENT", ""),
                      "loop_ms_by_position_in_segment": [sum(each[q::seg]) / len(each[q::seg]) * 1e3 for q in range(seg)]}))
else:
  print(f"back to back: {el / n * 1e3:.3f} ms per step; calls by position in the segment (ms): "
      + ", ".join(f"{sum(each[q::seg]) / len(each[q::seg]) * 1e3:.3f}" for q in range(seg)) + f"; slowest {max(each) * 1e3:.3f}")
