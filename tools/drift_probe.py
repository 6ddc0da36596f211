#!/usr/bin/env python3
"""Growth of the difference production vs bit-exact GPU path over N steps (c30 102x102x19), per tracer: tells a rounding-level
drift (smooth growth) from a flipped discrete decision (a jump at one step, e.g. a convective adjustment)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "oracle"))
import numpy as np
from uvic29_amd import synthetic
from uvic29_amd.tracer import TracerModel, TimeLoop

oc = synthetic.make_ocean("c30")
to, so, c = synthetic.load_eos(19)
ms = []
for exact in (True, False):
    m = TracerModel(102, 102, 19, oc.cfg.nt, oc.cfg.nsrc, oc.cfg.ntnpzd)
    m.set_exact(exact)
    m.load_ocean(oc, to, so, c)
    m.set_mobi(oc)
    ms.append((m, TimeLoop(m, oc.params.dtts, oc.params.nmix)))
names = oc.cfg.tracers
for step in range(1, 101):
    outs = []
    for m, loop in ms:
        loop.step()
    if step in (1, 2, 3, 5, 8, 12, 16, 17, 20, 30, 40, 50, 60, 70, 80, 90, 100):
        for m, loop in ms:
            m.sync()
            outs.append(m.download("t_tau"))
        d = [np.abs(outs[1][:, :, 1:101, n] - outs[0][:, :, 1:101, n]).max() / np.abs(outs[0][:, :, 1:101, n]).max() for n in range(len(names))]
        worst = int(np.argmax(d))
        # where is the largest temp difference
        dt = np.abs(outs[1][:, :, 1:101, 0] - outs[0][:, :, 1:101, 0])
        w = np.unravel_index(np.argmax(dt), dt.shape)
        print(f"step {step:3d}: temp {d[0]:.2e} salt {d[1]:.2e} worst {names[worst]} {d[worst]:.2e}  max|dT| at (i,k,j)=({w[0]+1},{w[1]+1},{w[2]+2})", flush=True)
