#!/usr/bin/env python3
"""Wall time of the Fortran overlay's `tracer` call (PCIe included) on c30 102x102x19, with the state uploaded and
downloaded every step or resident on the device (UVIC_RESIDENT=1).  One mode per process: the overlay reads the
environment once.    usage: [UVIC_RESIDENT=1] python tools/overlay_time.py [nsteps]"""
import os
import sys
import time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "oracle"))
from uvic29_amd import synthetic  # noqa: E402
import refdriver  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
oc = synthetic.make_ocean("c30", 102, 102, 19)
shim = refdriver.RefOcean(oc, shim=True)
shim.set_step_kind(False)
ts = []
for it in range(n):
    shim.isopyc(); shim.add_k33()          # host part of the "tracer only" integration, not timed
    t0 = time.perf_counter()
    shim.tracer()
    ts.append(time.perf_counter() - t0)
    shim.rotate()
ts = sorted(ts[2:])
g = oc.grid
units = g.imt * g.jmt * g.km * oc.cfg.nt
med = ts[len(ts) // 2]
print(f"overlay tracer call, {'resident' if os.environ.get('UVIC_RESIDENT') == '1' else 'upload/download every step'}: "
      f"median {med * 1e3:.2f} ms per step = {units / med / 1e9:.2f} G cell-updates/s (PCIe included)")
