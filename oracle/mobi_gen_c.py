"""ctypes binding of oracle/mobi_gen_oracle.c (MOBI with the option flags at run time) -- TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes

import numpy as np

import oracle_c
from mobi_c import Forcing, MAXK, MAXT

X = ("po4 phyt phyt_phos zoop detr detr_phos dic dic13 phytc13 zoopc13 detrc13 doc13 diazc13 "
     "dop no3 don diaz din15 don15 phytn15 zoopn15 detrn15 diazn15 dfe detrfe "
     "caco3 diat sil opl diatn15 diatc13 caco3c13").split()
SCALARS = ("kw kc ki tap abio_P bbio cbio nup nup_D nupt0 nupt0_D gamma1 gbio nuz nud0 nudon0 nudop0 "
           "redptn redctn redntp redotc redntc diazntp diazptn kzoo geZ zprefP zprefDet zprefZ zprefDiaz "
           "kfe_D kfemin kfemax knmin knmax pmax thetamaxlo thetamaxhi alphamin alphamax "
           "kfeleq kfeorg kfecol mc rfeton iscr jdiar dbct_D hdop dfr dfrt pfr "
           "eps_assim eps_recy eps_excr eps_nfix eps_wcdeni eps_bdeni0 capr").split()
SCALARS_CACO3 = "kc_c dissk0 caprmax kcapr".split()
SCALARS_SIL = "abiodiat kfemin_Diat kfemax_Diat knmin_Diat knmax_Diat pmax_Diat zprefDiat nu_diat nudt0 opl_disk0".split()
ARRAYS = ("wd", "ztt", "rcak", "rcab", "wc", "wo")


class OrcMobiG(ctypes.Structure):
    _fields_ = ([(n, ctypes.c_int) for n in ("km", "ntnpzd", "nsrc", "nbio", "opt_n15", "opt_c13", "opt_caco3", "opt_silicon")]
                + [(n, ctypes.c_double) for n in ("dtbio", "rdtts", "rnbio")]
                + [("im", ctypes.c_int * len(X)), ("is_", ctypes.c_int * len(X)), ("tracer_of_mobi", ctypes.c_int * MAXT)]
                + [(n, ctypes.c_int) for n in ("itemp", "isalt", "idic", "ialk", "io2", "ic14", "is_alk", "is_o2", "is_c14", "pad_")]
                + [(n, ctypes.c_double) for n in SCALARS + SCALARS_CACO3 + SCALARS_SIL]
                + [(n, ctypes.c_double * MAXK) for n in ARRAYS + ("zt", "dzt", "dztr")])


def supported(cfg) -> bool:
    o = cfg.options
    return all(x in o for x in ("mobi", "mobi_o2", "mobi_iron", "carbon", "mobi_alk", "mobi_nitrogen"))


def make_params(cfg, grid, prm: dict, c2dtts: float) -> OrcMobiG:
    """`prm`: COMMON /npzd_r/ after mobi_init (dict name -> float / list) incl. 'dtnpzd'."""
    if not supported(cfg):
        raise NotImplementedError(f"option set {cfg.name}: needs O_mobi_alk and O_mobi_nitrogen (set E reads t(..,ialk=0,..))")
    lib = oracle_c.lib()
    assert lib.orc_mobig_sizeof() == ctypes.sizeof(OrcMobiG), (lib.orc_mobig_sizeof(), ctypes.sizeof(OrcMobiG))
    P = OrcMobiG()
    km = grid.km
    P.km, P.ntnpzd, P.nsrc = km, cfg.ntnpzd, cfg.nsrc
    o = cfg.options
    P.opt_n15, P.opt_c13 = int("mobi_nitrogen_15" in o), int("carbon_13" in o)
    P.opt_caco3, P.opt_silicon = int("mobi_caco3" in o), int("mobi_silicon" in o)
    P.nbio = int(c2dtts / prm["dtnpzd"])          # tracer.F:340
    P.dtbio = c2dtts / P.nbio
    P.rdtts = 1. / c2dtts
    P.rnbio = 1. / P.nbio
    for q, n in enumerate(X):
        P.im[q] = cfg.imobi(n)
        P.is_[q] = cfg.sources.index(n) + 1 if n in cfg.sources else 0
    for m, name in enumerate(cfg.mobi):
        P.tracer_of_mobi[m] = cfg.index(name)
    P.itemp, P.isalt = cfg.index("temp"), cfg.index("salt")
    P.idic, P.ialk, P.io2, P.ic14 = cfg.index("dic"), cfg.index("alk"), cfg.index("o2"), cfg.index("c14")
    slot = lambda n: cfg.sources.index(n) + 1 if n in cfg.sources else 0
    P.is_alk, P.is_o2, P.is_c14 = slot("alk"), slot("o2"), slot("c14")
    names = SCALARS + (SCALARS_CACO3 if P.opt_caco3 else []) + (SCALARS_SIL if P.opt_silicon else [])
    for n in names:
        setattr(P, n, float(prm[n.lower()]))
    for n in ARRAYS:
        if (n == "wc" and not P.opt_caco3) or (n == "wo" and not P.opt_silicon):
            continue
        a = np.asarray(prm[n], dtype=np.float64)
        assert a.size == km, (n, a.size, km)
        for k in range(km):
            getattr(P, n)[k] = a[k]
    for k in range(km):
        P.zt[k], P.dzt[k], P.dztr[k] = grid.zt[k], grid.dzt[k], grid.dztr[k]
    return P


def mobi_sources(ocean, prm: dict, t_taum1, c2dtts):
    """src(imt,km,jmt,nsrc) for the whole synthetic ocean (tracer.F:311-545, 853-867)."""
    lib = oracle_c.lib()
    g, cfg = ocean.grid, ocean.cfg
    P = make_params(cfg, g, prm, c2dtts)
    F = Forcing(ocean)
    src = np.zeros((g.imt, g.km, g.jmt, cfg.nsrc), order="F")
    t = np.asfortranarray(t_taum1)
    kmt = np.asfortranarray(ocean.topo.kmt, dtype=np.int32)
    lib.orc_mobig_sources(ctypes.byref(P), ctypes.byref(F.c), g.imt, g.jmt, kmt.ctypes.data_as(ctypes.c_void_p),
                          t.ctypes.data_as(ctypes.c_void_p), cfg.nt, ctypes.c_double(c2dtts),
                          src.ctypes.data_as(ctypes.c_void_p))
    return src
