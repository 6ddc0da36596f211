#!/usr/bin/env python3
"""ms per step in consecutive blocks of steps from a cold start (how long the loop takes to reach its steady state)."""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from uvic29_amd import synthetic  # noqa: E402
from uvic29_amd.tracer import TimeLoop, TracerModel  # noqa: E402

blk = int(sys.argv[1]) if len(sys.argv) > 1 else 10
nblk = int(sys.argv[2]) if len(sys.argv) > 2 else 20
oc = synthetic.make_ocean("c30", 102, 102, 19)
m = TracerModel(102, 102, 19, oc.cfg.nt, oc.cfg.nsrc, oc.cfg.ntnpzd, device=0)
m.load_ocean(oc, *synthetic.load_eos(19))
m.set_mobi(oc)
loop = TimeLoop(m, oc.params.dtts, oc.params.nmix)
out = []
for b in range(nblk):
    m.sync()
    t0 = time.perf_counter()
    for _ in range(blk):
        loop.step()
    m.sync()
    out.append((time.perf_counter() - t0) / blk * 1e3)
print("ms/step per block of %d:" % blk, " ".join("%.3f" % x for x in out))
m.close()
