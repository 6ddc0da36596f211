import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle')
import numpy as np
from uvic29_amd import synthetic
from uvic29_amd.tracer import TracerModel, TimeLoop
oc = synthetic.make_ocean("c30")
to, so, c = synthetic.load_eos(19)
def run(exact, perturb=0.0, nsteps=100):
    m = TracerModel(102,102,19, oc.cfg.nt, oc.cfg.nsrc, oc.cfg.ntnpzd)
    m.set_exact(exact); m.load_ocean(oc, to, so, c); m.set_mobi(oc)
    if perturb:
        t = oc.t_tau.copy(order='F'); t *= (1.0 + perturb); m.upload('t_tau', t)   # one-ulp relative perturbation of t(tau)
    loop = TimeLoop(m, oc.params.dtts, oc.params.nmix)
    for _ in range(nsteps): loop.step()
    m.sync(); out = m.download('t_tau'); m.close(); return out
a = run(True); b = run(True, perturb=2.2e-16); f = run(False)
names = oc.cfg.tracers
rel = lambda x,y: np.abs(x-y)[:,:,1:101].max()/np.abs(y)[:,:,1:101].max()
print("tracer      exact-vs-1ulp-perturbed   fast-vs-exact")
for n,name in enumerate(names):
    print(f"{name:10s} {rel(b[...,n],a[...,n]):.2e}   {rel(f[...,n],a[...,n]):.2e}")
