"""ctypes binding of oracle/mobi_oracle.c -- TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes

import numpy as np

import oracle_c

MAXT, MAXK = 40, 64
_IDX = ("po4 phyt phyt_phos zoop detr detr_phos dic dic13 phytc13 zoopc13 detrc13 doc13 diazc13 "
        "dop no3 don diaz din15 don15 phytn15 zoopn15 detrn15 diazn15 dfe detrfe alk o2 c14").split()
SCALARS = ("kw kc ki tap abio_P bbio cbio nup nup_D nupt0 nupt0_D gamma1 gbio nuz nud0 nudon0 nudop0 "
           "redptn redctn redntp redotc redntc diazntp diazptn kzoo geZ zprefP zprefDet zprefZ zprefDiaz "
           "kfe_D kfemin kfemax knmin knmax pmax thetamaxlo thetamaxhi alphamin alphamax "
           "kfeleq kfeorg kfecol mc rfeton iscr jdiar dbct_D hdop dfr dfrt pfr "
           "eps_assim eps_recy eps_excr eps_nfix eps_wcdeni eps_bdeni0 capr").split()
ARRAYS = ("wd", "ztt", "rcak", "rcab")


class MobiIndex(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in _IDX]


class OrcMobi(ctypes.Structure):
    _fields_ = ([(n, ctypes.c_int) for n in ("km", "ntnpzd", "nsrc", "nbio")]
                + [(n, ctypes.c_double) for n in ("dtbio", "rdtts", "rnbio")]
                + [("im", MobiIndex), ("is_", MobiIndex), ("tracer_of_mobi", ctypes.c_int * MAXT)]
                + [(n, ctypes.c_int) for n in ("itemp", "isalt", "idic", "ialk", "io2", "ic14")]
                + [(n, ctypes.c_double) for n in SCALARS]
                + [(n, ctypes.c_double * MAXK) for n in ARRAYS + ("zt", "dzt", "dztr")])


_DP = ctypes.POINTER(ctypes.c_double)


class OrcForcing(ctypes.Structure):
    _fields_ = ([(n, ctypes.c_double) for n in ("pi", "radian", "relyr", "co2ccn")]
                + [(n, _DP) for n in ("tlat", "dnswr", "aice", "hice", "hsno", "sg_bathy", "fe_atmdep", "fe_hydr")])


def make_params(cfg, grid, prm: dict, c2dtts: float) -> OrcMobi:
    """`prm`: /npzd_r/ values after mobi_init (dict name -> float / list) incl. 'dtnpzd'."""
    P = OrcMobi()
    km = grid.km
    P.km, P.ntnpzd, P.nsrc = km, cfg.ntnpzd, cfg.nsrc
    P.nbio = int(c2dtts / prm["dtnpzd"])          # tracer.F:340
    P.dtbio = c2dtts / P.nbio
    P.rdtts = 1. / c2dtts
    P.rnbio = 1. / P.nbio
    for n in _IDX:
        setattr(P.im, n, cfg.imobi(n))
        setattr(P.is_, n, cfg.sources.index(n) + 1 if n in cfg.sources else 0)
    for m, name in enumerate(cfg.mobi):
        P.tracer_of_mobi[m] = cfg.index(name)
    P.itemp, P.isalt = cfg.index("temp"), cfg.index("salt")
    P.idic, P.ialk, P.io2, P.ic14 = cfg.index("dic"), cfg.index("alk"), cfg.index("o2"), cfg.index("c14")
    for n in SCALARS:
        setattr(P, n, float(prm[n.lower()]))
    for n in ARRAYS:
        a = np.asarray(prm[n], dtype=np.float64)
        assert a.size == km, (n, a.size, km)
        for k in range(km):
            getattr(P, n)[k] = a[k]
    for k in range(km):
        P.zt[k], P.dzt[k], P.dztr[k] = grid.zt[k], grid.dzt[k], grid.dztr[k]
    return P


class Forcing:
    def __init__(self, ocean):
        g, f, topo = ocean.grid, ocean.forcing, ocean.topo
        self.keep = {"tlat": np.asfortranarray(g.tlat), "dnswr": np.asfortranarray(f.dnswr),
                     "aice": np.asfortranarray(f.aice), "hice": np.asfortranarray(f.hice),
                     "hsno": np.asfortranarray(f.hsno), "sg_bathy": np.asfortranarray(topo.sg_bathy),
                     "fe_atmdep": np.asfortranarray(f.fe_atmdep), "fe_hydr": np.asfortranarray(f.fe_hydr)}
        self.c = OrcForcing()
        self.c.pi, self.c.radian, self.c.relyr, self.c.co2ccn = g.pi, g.radian, f.relyr, f.co2ccn
        for n, a in self.keep.items():
            setattr(self.c, n, a.ctypes.data_as(_DP))


def mobi_sources(ocean, prm: dict, t_taum1, c2dtts):
    """src(imt,km,jmt,nsrc) for the whole synthetic ocean (tracer.F:311-545, 853-867)."""
    lib = oracle_c.lib()
    g, cfg = ocean.grid, ocean.cfg
    P = make_params(cfg, g, prm, c2dtts)
    F = Forcing(ocean)
    src = np.zeros((g.imt, g.km, g.jmt, cfg.nsrc), order="F")
    t = np.asfortranarray(t_taum1)
    kmt = np.asfortranarray(ocean.topo.kmt, dtype=np.int32)
    lib.orc_mobi_sources(ctypes.byref(P), ctypes.byref(F.c), g.imt, g.jmt, kmt.ctypes.data_as(ctypes.c_void_p),
                         t.ctypes.data_as(ctypes.c_void_p), cfg.nt, ctypes.c_double(c2dtts),
                         src.ctypes.data_as(ctypes.c_void_p))
    return src
