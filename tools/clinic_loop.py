#!/usr/bin/env python3
"""state + clinic (polar filter and sbc accumulation on) N times on BASELINE's grid: the command profiled for
profiles/*_clinic_* (rocprofv3 --kernel-trace --stats -- python3 tools/clinic_loop.py 200)."""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from uvic29_amd import synthetic  # noqa: E402
from uvic29_amd.tracer import TracerModel  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
imt, jmt, km = (int(x) for x in (sys.argv[2].split("x") if len(sys.argv) > 2 else (102, 102, 19)))
oc = synthetic.make_ocean("m2", imt, jmt, km)
mom = synthetic.make_momentum(oc.grid, oc.topo, oc.u)
m = TracerModel(imt, jmt, km, oc.cfg.nt, oc.cfg.nsrc, oc.cfg.ntnpzd, device=0)
m.load_ocean(oc, *synthetic.load_eos(km))
m.load_momentum(oc, mom)
m.set_filter_u(oc, synthetic.make_filter_u(oc.grid, km))
m.state(); m.clinic_only(True)
m.sync()
t0 = time.perf_counter()
for _ in range(n):
    m.state_async(); m.clinic_async(True)
m.sync()
print(f"state+clinic {imt}x{jmt}x{km}: {(time.perf_counter() - t0) / n * 1e3:.4f} ms per call over {n} calls")
m.close()
