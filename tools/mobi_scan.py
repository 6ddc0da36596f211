#!/usr/bin/env python3
"""Time the MOBI source kernel against the number of Euler sub-steps (dtnpzd scan) to separate the
per-sub-step cost from the per-level prologue/epilogue.  GPU only; diagnostic, not a test."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from uvic29_amd import OPTION_SETS, synthetic, mobi as pm  # noqa: E402
from uvic29_amd.tracer import TracerModel  # noqa: E402

imt, jmt, km = 102, 102, 19
cfg = OPTION_SETS["c30"]
ocean = synthetic.make_ocean(cfg, imt, jmt, km)
to, so, c = synthetic.load_eos(km)
m = TracerModel(imt, jmt, km, cfg.nt, cfg.nsrc, cfg.ntnpzd, device=0)
m.load_ocean(ocean, to, so, c)
c2 = 2 * ocean.params.dtts
for nbio in (1, 2, 4, 8, 16):
    tab = dict(pm.load_table(cfg.name, km))
    tab["dtnpzd"] = c2 / nbio
    m.set_mobi(ocean, table=tab)
    p = m.profile(nrep=10)
    print(f"nbio={nbio:2d} mobi={p['mobi']:.4f} ms pre={p['mobi_pre']:.4f} post={p['mobi_post']:.4f} ms", flush=True)
