#!/usr/bin/env python3
"""Wall time of the Fortran overlay's `clinic` call (PCIe included) against the reference routine on one host core,
through the compiled reference's COMMON blocks (oracle/_ref build "m2", 102x102x19)."""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
for p in (ROOT, ROOT / "oracle"):
    sys.path.insert(0, str(p))
from uvic29_amd import synthetic  # noqa: E402
import refdriver  # noqa: E402

dims = (102, 102, 19)
oc = synthetic.make_ocean("m2", *dims)
mom = synthetic.make_momentum(oc.grid, oc.topo, oc.u)
out = {}
for shim in (False, True):
    R = refdriver.RefOcean(oc, shim=shim)
    R.set_momentum(mom)
    R.set_filter(synthetic.make_filter(oc.grid, dims[2]))
    R.set_filter_u(synthetic.make_filter_u(oc.grid, dims[2]))
    R.state(); R.adv_vel_u(); R.setvbc()
    if shim:
        R.step()                 # the tracer overlay makes the device instance
    g = oc.grid
    for _ in range(3):
        R.ref.call("clinic", 0, 2, g.jmt - 1, 2, g.imt - 1)
    n = 20
    t0 = time.perf_counter()
    for it in range(n):
        R.ref.set("itt", it + 100)       # the overlay sends the advective velocities itself on every call
        R.ref.call("clinic", 0, 2, g.jmt - 1, 2, g.imt - 1)
    out["overlay" if shim else "reference"] = (time.perf_counter() - t0) / n * 1e3
print("clinic 102x102x19, ms per call: reference (1 core) %.3f, overlay (PCIe included) %.3f" % (out["reference"], out["overlay"]))
