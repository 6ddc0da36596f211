#!/usr/bin/env python3
"""What one ocean step costs on the HOST in the compiled reference (oracle/_ref, one core), routine by routine,
on the 102x102x19 grid: which of mom.F's calls (source/mom/mom.F:300-420) bound the model once `tracer` and
`clinic` are served by the device.  CPU only; needs oracle/_ref (built by __graft_entry__.build() where
/root/reference exists).  Test infrastructure: nothing here is on the product path.

    python tools/host_costs.py
"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
import numpy as np

from uvic29_amd import synthetic
import refdriver


def main():
    oc = synthetic.make_ocean("m2", 102, 102, 19)
    mom = synthetic.make_momentum(oc.grid, oc.topo, oc.u, anisotropic=True)
    R = refdriver.RefOcean(oc)
    R.set_momentum(mom)
    g = oc.grid

    def ms(f, n=3):
        f()
        t0 = time.perf_counter()
        for _ in range(n):
            f()
        return (time.perf_counter() - t0) / n * 1e3

    tid = synthetic.make_tidal(g, oc.topo)
    bg = np.asfortranarray(oc.diff_cbt_bg)
    rows = [
        ("state", lambda: R.state()),
        ("adv_vel", lambda: R.ref.call("adv_vel", 0, 1, g.jmt, 2, g.imt - 1)),
        ("isopyc", lambda: R.isopyc()),
        ("vmixc (O_tidal_kv)", lambda: R.vmixc(tid, bg)),
        ("setvbc", lambda: R.setvbc()),
        ("clinic", lambda: R.clinic()),
        ("tracer, T and S only", lambda: R.tracer()),
    ]
    for name, f in rows:
        print("%-24s %7.1f ms" % (name, ms(f)))


if __name__ == "__main__":
    main()
