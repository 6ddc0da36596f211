#!/usr/bin/env python3
"""Run the MOBI pass once with the -DUV_MOBI_TIMING build (per-phase cycle counters printed by the team kernel)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from uvic29_amd import OPTION_SETS, synthetic  # noqa: E402
from uvic29_amd.tracer import TracerModel  # noqa: E402
imt, jmt, km = 102, 102, 19
cfg = OPTION_SETS["c30"]
ocean = synthetic.make_ocean(cfg, imt, jmt, km)
to, so, c = synthetic.load_eos(km)
m = TracerModel(imt, jmt, km, cfg.nt, cfg.nsrc, cfg.ntnpzd, device=0)
m.load_ocean(ocean, to, so, c)
m.set_mobi(ocean)
m.mobi(); m.sync()
m.mobi(); m.sync()
