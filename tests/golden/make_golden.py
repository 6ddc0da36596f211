#!/usr/bin/env python3
"""Generate the committed golden fixtures from the reference itself.

Run in the build container only (needs /root/reference and oracle/_ref built by
oracle/build_ref.py).  The reference has no tests or golden vectors of its own
(SURVEY.md §4), so every vector here is an OUTPUT OF THE COMPILED REFERENCE on
the deterministic synthetic ocean of uvic2.9_amd/synthetic.py; inputs are
regenerated from that module by the tests, so only outputs are stored.

Fixtures
  uvic2.9_amd/data/eos.json            to, so, c(km,9) of the reference's eqstate
                                       (source/mom/denscoef.F) for the synthetic
                                       vertical grids (inputs of the hot path)
  tests/golden/step_<cfg>_<grid>.npz   t(tau+1) after one isopyc+tracer step,
                                       K33 and the GM velocities
  tests/golden/run_<cfg>_<grid>.npz    t after N leapfrog steps (mixing step every
                                       nmix-th) and the tbar/travar integrals
  tests/golden/clinic_m2_<grid>.npz    rho of the reference's state; grad_p, zu and u(tau+1) of its clinic, without
                                       and with the polar filter filuv (momentum step)
  tests/golden/prep_<cfg>_<grid>.npz   outputs of the reference's adv_vel, vmixc (tidal
                                       mixing + K33 after isopyc), findex and of one
                                       isopyc+tracer step with the polar filter on
"""
from __future__ import annotations

import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))

from uvic29_amd import synthetic  # noqa: E402
import refdriver  # noqa: E402
import refmodel  # noqa: E402

HERE = Path(__file__).resolve().parent


def eos_fixture():
    ref = refmodel.RefLib("p2", 14, 14, 6)
    out = {}
    for km in (6, 19, 32):
        g = synthetic.make_grid(14, 14, km)
        zt = np.ascontiguousarray(g.zt)
        ro0, to, so = np.zeros(km), np.zeros(km), np.zeros(km)
        c = np.zeros((km, 9), order="F")
        tmink, tmaxk, smink, smaxk = (np.zeros(km) for _ in range(4))
        ref.call("eqstate", zt, km, ro0, to, so, c, tmink, tmaxk, smink, smaxk)
        out[str(km)] = {"zt": zt.tolist(), "to": to.tolist(), "so": so.tolist(),
                        "c": [c[:, m].tolist() for m in range(9)]}
    p = ROOT / "uvic2.9_amd" / "data" / "eos.json"
    p.parent.mkdir(exist_ok=True)
    p.write_text(json.dumps(out))
    print("wrote", p)


MOBI_SCALARS = ("kw kc ki tap abio_p bbio cbio nup nup_d nupt0 nupt0_d gamma1 gbio nuz nud0 nudon0 nudop0 "
                "redptn redctn redntp redotc redntc diazntp diazptn kzoo gez zprefp zprefdet zprefz zprefdiaz "
                "kfe_d kfemin kfemax knmin knmax pmax thetamaxlo thetamaxhi alphamin alphamax "
                "kfeleq kfeorg kfecol mc rfeton iscr jdiar dbct_d hdop dfr dfrt pfr "
                "eps_assim eps_recy eps_excr eps_nfix eps_wcdeni eps_bdeni0 capr dtnpzd").split()


def mobi_fixture():
    """COMMON /npzd_r/ after the reference's own mobi_init (run/control.in, silicon-only
    namelist members removed) for every synthetic vertical grid."""
    import build_ref
    out = {}
    for km in (6, 19, 32):
        if not refmodel.available("c30", 14, 14, km):
            build_ref.build("c30", 14, 14, km)
        oc = synthetic.make_ocean("c30", 14, 14, km)
        ro = refdriver.RefOcean(oc)
        d = {n: float(ro.v[n][0]) for n in MOBI_SCALARS}
        for n in ("wd", "ztt", "rcak", "rcab"):
            d[n] = ro.v[n].tolist()
        out[str(km)] = d
    p = ROOT / "uvic2.9_amd" / "data" / "mobi_c30.json"
    p.write_text(json.dumps(out))
    print("wrote", p)


def mobi_set_fixture(cfg_name):
    """The same for the option sets served by the run-time-flag MOBI path (f18, s37): every scalar and level array of
    COMMON /npzd_r/ the path reads, after the reference's own mobi_init, for km = 6 and 19."""
    import build_ref
    import mobi_gen_c
    out = {}
    for km in (6, 19):
        if not refmodel.available(cfg_name, 14, 14, km):
            build_ref.build(cfg_name, 14, 14, km)
        oc = synthetic.make_ocean(cfg_name, 14, 14, km)
        ro = refdriver.RefOcean(oc)
        o = oc.cfg.options
        names = (mobi_gen_c.SCALARS + (mobi_gen_c.SCALARS_CACO3 if "mobi_caco3" in o else [])
                 + (mobi_gen_c.SCALARS_SIL if "mobi_silicon" in o else []) + ["dtnpzd"])
        d = {n.lower(): float(ro.v[n.lower()][0]) for n in names}
        for n in mobi_gen_c.ARRAYS:
            if (n == "wc" and "mobi_caco3" not in o) or (n == "wo" and "mobi_silicon" not in o):
                continue
            d[n] = ro.v[n].tolist()
        # the column order of tnpzd as tracer_init assigned it
        d["imobi"] = {n: int(ro.v["imobi" + n][0]) for n in mobi_gen_c.X if "imobi" + n in ro.v and oc.cfg.imobi(n)}
        out[str(km)] = d
    p = ROOT / "uvic2.9_amd" / "data" / f"mobi_{cfg_name}.json"
    p.write_text(json.dumps(out))
    print("wrote", p)


def step_fixture(cfg, imt, jmt, km):
    oc = synthetic.make_ocean(cfg, imt, jmt, km)
    ro = refdriver.RefOcean(oc)
    tp = ro.step().copy()
    v = ro.v
    np.savez_compressed(HERE / f"step_{cfg}_{imt}x{jmt}x{km}.npz", t_taup1=tp, k33=v["k33"].copy(),
                        adv_vetiso=v["adv_vetiso"].copy(), adv_vntiso=v["adv_vntiso"].copy(),
                        adv_vbtiso=v["adv_vbtiso"].copy())
    print("wrote step", cfg, imt, jmt, km)


def run_fixture(cfg, imt, jmt, km, nsteps):
    oc = synthetic.make_ocean(cfg, imt, jmt, km)
    ro = refdriver.RefOcean(oc)
    dtts, nmix = oc.params.dtts, oc.params.nmix
    for it in range(1, nsteps + 1):
        mixing = (it % nmix) == 0
        if mixing:                      # forward step: both slots hold tau (loadmw.F:107-111)
            ro.v["t"][..., 0] = ro.v["t"][..., 1]
        ro.step(c2dtts=dtts if mixing else 2.0 * dtts)
        ro.rotate()
    t = ro.v["t"][..., 1].copy()
    np.savez_compressed(HERE / f"run_{cfg}_{imt}x{jmt}x{km}_n{nsteps}.npz", t=t)
    print("wrote run", cfg, imt, jmt, km, nsteps)


def prep_fixture(cfg, imt, jmt, km):
    oc = synthetic.make_ocean(cfg, imt, jmt, km)
    tid = synthetic.make_tidal(oc.grid, oc.topo, oc.params.kappa_h)
    flt = synthetic.make_filter(oc.grid, km)
    ro = refdriver.RefOcean(oc)
    vet, vnt, vbt = ro.adv_vel()
    ro.isopyc()
    dcb = ro.vmixc(tid, np.asfortranarray(oc.diff_cbt_bg))
    ro = refdriver.RefOcean(oc)
    istf, ietf = ro.set_filter(flt)
    tp = ro.step().copy()
    np.savez_compressed(HERE / f"prep_{cfg}_{imt}x{jmt}x{km}.npz", adv_vet=vet, adv_vnt=vnt, adv_vbt=vbt, diff_cbt=dcb,
                        istf=istf, ietf=ietf, t_taup1_filtered=tp)
    print("wrote prep", cfg, imt, jmt, km)


def clinic_fixture(imt, jmt, km):
    """Outputs of the reference's state, clinic (with and without filuv) on the synthetic ocean, configuration "m2"."""
    oc = synthetic.make_ocean("m2", imt, jmt, km)
    mom = synthetic.make_momentum(oc.grid, oc.topo, oc.u)
    flt = synthetic.make_filter_u(oc.grid, km)
    ro = refdriver.RefOcean(oc)
    ro.set_momentum(mom)
    rho = ro.state()
    ro.adv_vel_u()
    ro.setvbc()
    up, zu, gp = ro.clinic()
    ro.set_filter_u(flt)
    upf, _, _ = ro.clinic()
    np.savez_compressed(HERE / f"clinic_m2_{imt}x{jmt}x{km}.npz", rho=rho, grad_p=gp, zu=zu, u_taup1=up, u_taup1_filtered=upf)
    print("wrote clinic", imt, jmt, km, "filter changed", int((up != upf).sum()), "values")


if __name__ == "__main__":
    clinic_fixture(14, 14, 6)
    prep_fixture("p2", 14, 14, 6)
    eos_fixture()
    mobi_fixture()
    step_fixture("p2", 14, 14, 6)
    step_fixture("c30", 14, 14, 6)
    run_fixture("p2", 14, 14, 6, 20)
    run_fixture("c30", 14, 14, 6, 20)
    for name in ("f18", "s37"):
        mobi_set_fixture(name)
        step_fixture(name, 14, 14, 6)
