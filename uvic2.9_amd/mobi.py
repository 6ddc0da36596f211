"""MOBI parameters and forcing on the host side of the C ABI.

`MobiParams` mirrors `uvic_mobi_params` (include/uvic_gpu.h): the contents of
COMMON /npzd_r/ after `mobi_init` plus the index maps.  In the real model the
Fortran shim fills it from COMMON; for the synthetic benchmark the values come
from data/mobi_c30.json, which tests/golden/make_golden.py wrote from the
compiled reference's own `mobi_init` run on run/control.in.
"""
from __future__ import annotations

import ctypes
import json
from pathlib import Path

import numpy as np

_IDX = ("po4 phyt phyt_phos zoop detr detr_phos dic dic13 phytc13 zoopc13 detrc13 doc13 diazc13 "
        "dop no3 don diaz din15 don15 phytn15 zoopn15 detrn15 diazn15 dfe detrfe alk o2 c14").split()
SCALARS = ("kw kc ki tap abio_P bbio cbio nup nup_D nupt0 nupt0_D gamma1 gbio nuz nud0 nudon0 nudop0 "
           "redptn redctn redntp redotc redntc diazntp diazptn kzoo geZ zprefP zprefDet zprefZ zprefDiaz "
           "kfe_D kfemin kfemax knmin knmax pmax thetamaxlo thetamaxhi alphamin alphamax "
           "kfeleq kfeorg kfecol mc rfeton iscr jdiar dbct_D hdop dfr dfrt pfr "
           "eps_assim eps_recy eps_excr eps_nfix eps_wcdeni eps_bdeni0 capr").split()
ARRAYS = ("wd", "ztt", "rcak", "rcab")


class MobiIndex(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in _IDX]


class MobiParams(ctypes.Structure):
    _fields_ = ([(n, ctypes.c_int32) for n in ("km", "ntnpzd", "nsrc", "pad_")]
                + [("im", MobiIndex), ("is_", MobiIndex), ("tracer_of_mobi", ctypes.c_int32 * 40),
                   ("slot_of_mobi", ctypes.c_int32 * 40)]
                + [(n, ctypes.c_int32) for n in ("itemp", "isalt", "idic", "ialk", "io2", "ic14")]
                + [("dtnpzd", ctypes.c_double)]
                + [(n, ctypes.c_double) for n in SCALARS]
                + [(n, ctypes.c_double * 64) for n in ARRAYS + ("zt", "dzt", "dztr")])


_DP = ctypes.POINTER(ctypes.c_double)


# every MOBI column tracer of every option set, in the order of uvic_mobi_options.im / .is (include/uvic_gpu.h)
X = ("po4 phyt phyt_phos zoop detr detr_phos dic dic13 phytc13 zoopc13 detrc13 doc13 diazc13 "
     "dop no3 don diaz din15 don15 phytn15 zoopn15 detrn15 diazn15 dfe detrfe "
     "caco3 diat sil opl diatn15 diatc13 caco3c13").split()
OPT_SCALARS = ("kc_c dissk0 caprmax kcapr abiodiat kfemin_Diat kfemax_Diat knmin_Diat knmax_Diat pmax_Diat "
               "zprefDiat nu_diat nudt0 opl_disk0").split()


class MobiOptions(ctypes.Structure):
    _fields_ = ([(n, ctypes.c_int32) for n in ("n15", "c13", "caco3", "silicon")]
                + [("im", ctypes.c_int32 * len(X)), ("is_", ctypes.c_int32 * len(X))]
                + [(n, ctypes.c_int32) for n in ("is_alk", "is_o2", "is_c14", "pad_")]
                + [(n, ctypes.c_double) for n in OPT_SCALARS]
                + [("wc", ctypes.c_double * 64), ("wo", ctypes.c_double * 64)])


def supported(cfg) -> bool:
    """Option sets the device serves: everything the reference can build with a defined result --
    O_mobi_alk and O_mobi_nitrogen on; set E is out (the reference reads t(..,ialk=0,..) there)."""
    o = cfg.options
    return all(x in o for x in ("mobi", "mobi_o2", "mobi_iron", "carbon", "mobi_alk", "mobi_nitrogen"))


def is_set_c(cfg) -> bool:
    o = cfg.options
    return "mobi_nitrogen_15" in o and "carbon_13" in o and "mobi_caco3" not in o and "mobi_silicon" not in o


def make_options(cfg, grid, prm: dict) -> MobiOptions:
    o = cfg.options
    O = MobiOptions()
    O.n15, O.c13 = int("mobi_nitrogen_15" in o), int("carbon_13" in o)
    O.caco3, O.silicon = int("mobi_caco3" in o), int("mobi_silicon" in o)
    slot = lambda n: cfg.sources.index(n) + 1 if n in cfg.sources else 0  # noqa: E731
    for q, n in enumerate(X):
        O.im[q], O.is_[q] = cfg.imobi(n), slot(n)
    O.is_alk, O.is_o2, O.is_c14 = slot("alk"), slot("o2"), slot("c14")
    names = (OPT_SCALARS[:4] if O.caco3 else []) + (OPT_SCALARS[4:] if O.silicon else [])
    for n in names:
        setattr(O, n, float(prm[n.lower()]))
    for n, on in (("wc", O.caco3), ("wo", O.silicon)):
        if on:
            a = np.asarray(prm[n], dtype=np.float64)
            if a.size != grid.km:
                raise ValueError(f"MOBI array {n} has {a.size} levels, model has {grid.km}")
            for k in range(grid.km):
                getattr(O, n)[k] = a[k]
    return O


class MobiForcing(ctypes.Structure):
    _fields_ = ([(n, ctypes.c_double) for n in ("pi", "radian", "relyr", "co2ccn")]
                + [(n, _DP) for n in ("tlat", "dnswr", "aice", "hice", "hsno", "sg_bathy", "fe_atmdep", "fe_hydr")])


def load_table(cfg_name: str, km: int) -> dict:
    cfg_name = {"t30": "c30"}.get(cfg_name, cfg_name)     # (t30: option set C in the oracle/_ref build that also holds `clinic` and the time-step monitor)
    tab = json.loads((Path(__file__).resolve().parent / "data" / f"mobi_{cfg_name}.json").read_text())
    if str(km) not in tab:
        raise KeyError(f"no MOBI parameter table for {cfg_name} with km={km}; available: {sorted(tab)}")
    return tab[str(km)]


def make_params(cfg, grid, prm: dict) -> MobiParams:
    if not supported(cfg):
        raise NotImplementedError(f"option set {cfg.name}: the device serves the sets with O_mobi_alk and O_mobi_nitrogen "
                                  "(C, F, the shipped nt=37 set); set E is undefined in the reference itself (ialk = 0)")
    P = MobiParams()
    km = grid.km
    P.km, P.ntnpzd, P.nsrc = km, cfg.ntnpzd, cfg.nsrc
    for n in _IDX:
        setattr(P.im, n, cfg.imobi(n) if n in cfg.mobi else 0)
        setattr(P.is_, n, cfg.sources.index(n) + 1 if n in cfg.sources else 0)
    for m, name in enumerate(cfg.mobi):
        P.tracer_of_mobi[m] = cfg.index(name)
        P.slot_of_mobi[m] = cfg.sources.index(name) + 1
    P.itemp, P.isalt = cfg.index("temp"), cfg.index("salt")
    P.idic, P.ialk, P.io2, P.ic14 = cfg.index("dic"), cfg.index("alk"), cfg.index("o2"), cfg.index("c14")
    P.dtnpzd = float(prm["dtnpzd"])
    for n in SCALARS:
        setattr(P, n, float(prm[n.lower()]))
    for n in ARRAYS:
        a = np.asarray(prm[n], dtype=np.float64)
        if a.size != km:
            raise ValueError(f"MOBI array {n} has {a.size} levels, model has {km}")
        for k in range(km):
            getattr(P, n)[k] = a[k]
    for k in range(km):
        P.zt[k], P.dzt[k], P.dztr[k] = grid.zt[k], grid.dzt[k], grid.dztr[k]
    return P


class Forcing:
    """Keeps the host arrays alive for the duration of the call."""

    def __init__(self, ocean):
        g, f, topo = ocean.grid, ocean.forcing, ocean.topo
        self.keep = {"tlat": np.asfortranarray(g.tlat), "dnswr": np.asfortranarray(f.dnswr),
                     "aice": np.asfortranarray(f.aice), "hice": np.asfortranarray(f.hice),
                     "hsno": np.asfortranarray(f.hsno), "sg_bathy": np.asfortranarray(topo.sg_bathy),
                     "fe_atmdep": np.asfortranarray(f.fe_atmdep), "fe_hydr": np.asfortranarray(f.fe_hydr)}
        self.c = MobiForcing()
        self.c.pi, self.c.radian, self.c.relyr, self.c.co2ccn = g.pi, g.radian, f.relyr, f.co2ccn
        for n, a in self.keep.items():
            setattr(self.c, n, a.ctypes.data_as(_DP))
