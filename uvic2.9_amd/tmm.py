"""Host-side mirror of the reference's second boundary: `tracer` compiled with -DO_TMM, the column-batch source
operator the Transport-Matrix-Method driver calls (updates/09/source/mom/tracer.F:109-124,
updates/09/source/common/size.h:26-30: imt = batch size, jmt = 1).  Columns in, source terms out:

    op = TmmOperator(cfg, grid, ncols)                      # cfg: OptionSet, grid: vertical grid (km, zt, dzt, ...)
    op.set_columns(kmt, tlat, sg_bathy, fe_atmdep, fe_hydr, dnswr, aice, hice, hsno, relyr, co2ccn)
    src = op.sources(t_taum1, c2dtts)                       # (ncols, km, nt) -> (ncols, km, nsrc)

Everything runs in libuvic_gpu.so (uvic_gpu_tmm_*); there is no CPU path.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import mobi as pm
from .capi import UvicGpuError, check, load

_DP = ctypes.POINTER(ctypes.c_double)


class TmmOperator:
    def __init__(self, cfg, grid, ncols: int, device: int = 0, table: dict | None = None):
        if not cfg.ntnpzd:
            raise UvicGpuError("the O_TMM operator is the MOBI source loop: the option set has no MOBI tracers")
        self.lib = load()
        self.cfg, self.grid, self.ncols = cfg, grid, int(ncols)
        self.km, self.nt, self.nsrc = grid.km, cfg.nt, cfg.nsrc
        self.table = table if table is not None else pm.load_table(cfg.name, grid.km)
        self.h = ctypes.c_void_p()
        check(self.lib.uvic_gpu_tmm_create(ctypes.byref(self.h), self.ncols, self.km, self.nt, self.nsrc, cfg.ntnpzd, device),
              "tmm_create")
        self.relyr = self.co2ccn = 0.0

    def set_columns(self, kmt, tlat, sg_bathy, fe_atmdep, fe_hydr, dnswr, aice, hice, hsno, relyr, co2ccn):
        """kmt, tlat, dnswr, aice, hice, hsno: (ncols); sg_bathy, fe_hydr: (ncols, km); fe_atmdep: (ncols, 12)."""
        g, nc = self.grid, self.ncols
        f64 = lambda a, shape: np.asfortranarray(np.asarray(a, dtype=np.float64).reshape(shape))
        keep = {"tlat": f64(tlat, (nc,)), "dnswr": f64(dnswr, (nc,)), "aice": f64(aice, (nc,)), "hice": f64(hice, (nc,)),
                "hsno": f64(hsno, (nc,)), "sg_bathy": f64(sg_bathy, (nc, self.km)), "fe_atmdep": f64(fe_atmdep, (nc, 12)),
                "fe_hydr": f64(fe_hydr, (nc, self.km))}
        F = pm.MobiForcing()
        F.pi, F.radian, F.relyr, F.co2ccn = g.pi, g.radian, relyr, co2ccn
        for n, a in keep.items():
            setattr(F, n, a.ctypes.data_as(_DP))
        k = np.ascontiguousarray(kmt, dtype=np.int32)
        P = pm.make_params(self.cfg, g, self.table)
        O = None if pm.is_set_c(self.cfg) else pm.make_options(self.cfg, g, self.table)
        check(self.lib.uvic_gpu_tmm_set_mobi(self.h, k.ctypes.data_as(ctypes.c_void_p), ctypes.byref(P),
                                             ctypes.byref(O) if O is not None else None, ctypes.byref(F)), "tmm_set_mobi")
        self.relyr, self.co2ccn = relyr, co2ccn

    def sources(self, t_taum1, c2dtts, relyr=None, co2ccn=None, forcing=None):
        """t_taum1 (ncols, km, nt) -> src (ncols, km, nsrc).  forcing: None (unchanged) or (dnswr, aice, hice, hsno)."""
        t = np.asfortranarray(t_taum1, dtype=np.float64)
        if t.shape != (self.ncols, self.km, self.nt):
            raise UvicGpuError(f"sources: t_taum1 has shape {t.shape}, expected {(self.ncols, self.km, self.nt)}")
        src = np.zeros((self.ncols, self.km, self.nsrc), order="F")
        relyr = self.relyr if relyr is None else relyr
        co2ccn = self.co2ccn if co2ccn is None else co2ccn
        fp = [None] * 4
        if forcing is not None:
            keep = [np.ascontiguousarray(a, dtype=np.float64) for a in forcing]
            fp = [a.ctypes.data_as(_DP) for a in keep]
        check(self.lib.uvic_gpu_tmm_sources(self.h, float(c2dtts), float(relyr), float(co2ccn), t.ctypes.data_as(_DP), *fp,
                                            src.ctypes.data_as(_DP)), "tmm_sources")
        self.relyr, self.co2ccn = relyr, co2ccn
        return src

    def close(self):
        if self.h:
            self.lib.uvic_gpu_destroy(self.h)
            self.h = ctypes.c_void_p()
