#!/usr/bin/env python3
"""Shader clock held by the MOBI team kernel alone and inside the full time loop (build with -DUV_CLOCK_PROBE, run with
UVIC_GPU_LIB pointing at that build): the kernel prints d(s_memtime)/d(s_memrealtime)."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "oracle"))
from uvic29_amd import synthetic
from uvic29_amd.tracer import TracerModel, TimeLoop
oc = synthetic.make_ocean("c30")
to, so, c = synthetic.load_eos(19)
m = TracerModel(102, 102, 19, oc.cfg.nt, oc.cfg.nsrc, oc.cfg.ntnpzd)
m.load_ocean(oc, to, so, c)
m.set_mobi(oc)
print("--- MOBI alone (uvic_gpu_mobi x 20) ---", flush=True)
for _ in range(20):
    m.mobi()
m.sync()
print("--- in the time loop (200 steps) ---", flush=True)
loop = TimeLoop(m, oc.params.dtts, oc.params.nmix)
for _ in range(200):
    loop.step()
m.sync()
