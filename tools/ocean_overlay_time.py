#!/usr/bin/env python3
"""Wall time per time step of the two Fortran overlays together -- `tracer` then `clinic`, as source/mom/mom.F:389-395
calls them -- on the 102x102x19 grid (oracle/_ref build "m2": T and S only), PCIe included, with the velocities shipped on
every call (UVIC_RESIDENT=1) or resident on the device (UVIC_RESIDENT=2: psi and the wind stress up, zu down).  The host
routines of the loop (loadmw's add_ext_mode, state, adv_vel, isopyc, setvbc) are run but not timed.
usage: UVIC_RESIDENT=1|2 python tools/ocean_overlay_time.py [nsteps]"""
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

n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
dims = (102, 102, 19)
oc = synthetic.make_ocean("m2", *dims)
mom = synthetic.make_momentum(oc.grid, oc.topo, oc.u)
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
tt, tc = [], []
for it in range(1, n + 1):
    S("itt", it)
    R.add_ext_mode(_psi(g, it), "tau")
    if it == 1:
        R.add_ext_mode(_psi(g, 0), "tau-1")
    R.state()
    R.ref.call("adv_vel", 0, 1, g.jmt, 2, g.imt - 1)
    R.isopyc(); R.add_k33(); R.setvbc()
    t0 = time.perf_counter()
    R.ref.call("tracer", 0, 2, g.jmt - 1, 2, g.imt - 1)
    t1 = time.perf_counter()
    R.ref.call("clinic", 0, 2, g.jmt - 1, 2, g.imt - 1)
    t2 = time.perf_counter()
    tt.append(t1 - t0); tc.append(t2 - t1)
    R.rotate()
    u = v["u"]; u[..., 0] = u[..., 1]; u[..., 1] = u[..., 2]
med = lambda x: sorted(x[4:])[len(x[4:]) // 2] * 1e3
print(f"UVIC_RESIDENT={os.environ.get('UVIC_RESIDENT', '')}: tracer call {med(tt):.3f} ms, clinic call {med(tc):.3f} ms (medians over {n - 4} steps, PCIe included)")
