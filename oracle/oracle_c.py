"""ctypes binding of the CPU restatement (oracle/uvic_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package.
"""
from __future__ import annotations

import ctypes
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
LIB = HERE / "liboracle.so"
SOURCES = [HERE / "uvic_oracle.c", HERE / "mobi_oracle.c", HERE / "mobi_gen_oracle.c", HERE / "prep_oracle.c",
           HERE / "filter_oracle.c", HERE / "clinic_oracle.c"]

_D = ctypes.POINTER(ctypes.c_double)
_I = ctypes.POINTER(ctypes.c_int)

_INTS = ["imt", "jmt", "km", "nt", "nsrc"]
_SCALARS = ["c2dtts", "aidif", "diff_cet", "diff_cnt", "slmxr", "ahisop", "athkdf"]
_DPTR = ["dxt", "dxtr", "dxu", "dxur", "dxt4r", "dyt", "dytr", "dyu", "dyur", "dyt4r",
         "cst", "cstr", "csu", "cstdytr", "cstdyt2r", "csu_dyur",
         "dzt", "dztr", "dzt2r", "dztur", "dztlr", "dzw", "dzwr", "dtxcel", "dtxsqr", "dztxcl",
         "to", "so", "c"]
_AFTER_KMT = ["tmask", "fisop", "addisop", "t_taum1", "t_tau", "t_taup1", "adv_vet", "adv_vnt", "adv_vbt",
              "diff_cbt", "stf", "btf", "src"]
_TAIL = ["alphai", "betai", "ddxt", "ddyt", "ddzt", "Ai_ez", "Ai_nz", "Ai_bx", "Ai_by", "K11", "K22", "K33",
         "adv_vetiso", "adv_vntiso", "adv_vbtiso", "adv_fe", "adv_fn", "adv_fb", "diff_fe", "diff_fn",
         "diff_fb", "diff_fbiso"]


class OrcCtx(ctypes.Structure):
    _fields_ = ([(n, ctypes.c_int) for n in _INTS] + [(n, ctypes.c_double) for n in _SCALARS]
                + [(n, _D) for n in _DPTR] + [("kmt", _I)] + [(n, _D) for n in _AFTER_KMT]
                + [("itrc", _I)] + [(n, _D) for n in _TAIL])


def build(force: bool = False) -> Path:
    srcs = [s for s in SOURCES if s.exists()]
    if LIB.exists() and not force and all(LIB.stat().st_mtime >= s.stat().st_mtime for s in srcs + list(HERE.glob("*.h"))):
        return LIB
    cmd = ["gcc", "-O2", "-ffp-contract=off", "-fopenmp", "-fPIC", "-shared", "-std=gnu99", "-o", str(LIB)] + [str(s) for s in srcs] + ["-lm"]
    subprocess.run(cmd, check=True)
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(str(LIB))
    return _lib


def _f(shape):
    return np.zeros(shape, dtype=np.float64, order="F")


class Oracle:
    """Holds an orc_ctx built from a synthetic Ocean; arrays are numpy, F order."""

    def __init__(self, ocean, to=None, so=None, c=None, src=None):
        g, topo, cfg, prm = ocean.grid, ocean.topo, ocean.cfg, ocean.params
        imt, jmt, km, nt = g.imt, g.jmt, g.km, cfg.nt
        self.ocean = ocean
        self.a = a = {}
        for n in _DPTR[:-3]:
            a[n] = np.ascontiguousarray(getattr(g, n), dtype=np.float64)
        a["to"] = np.ascontiguousarray(to, dtype=np.float64)
        a["so"] = np.ascontiguousarray(so, dtype=np.float64)
        a["c"] = np.asfortranarray(c, dtype=np.float64)
        a["kmt"] = np.asfortranarray(topo.kmt, dtype=np.int32)
        a["tmask"] = np.asfortranarray(topo.tmask)
        a["fisop"] = np.asfortranarray(ocean.fisop)
        a["addisop"] = np.asfortranarray(ocean.addisop)
        a["t_taum1"] = np.array(ocean.t_taum1, order="F")
        a["t_tau"] = np.array(ocean.t_tau, order="F")
        a["t_taup1"] = _f((imt, km, jmt, nt))
        a["adv_vet"], a["adv_vnt"], a["adv_vbt"] = ocean.adv_vet, ocean.adv_vnt, ocean.adv_vbt
        a["diff_cbt"] = _f((imt, km, jmt))
        a["stf"], a["btf"] = ocean.stf, ocean.btf
        a["src"] = None if src is None else np.asfortranarray(src)
        a["itrc"] = np.array(cfg.itrc(), dtype=np.int32)
        for n in ("alphai", "betai", "K11", "K22", "K33", "adv_vetiso", "adv_vntiso", "adv_fe", "adv_fn",
                  "diff_fe", "diff_fn"):
            a[n] = _f((imt, km, jmt))
        for n in ("ddxt", "ddyt"):
            a[n] = _f((imt, km, jmt, 2))
        a["ddzt"] = _f((imt, km + 1, jmt, 2))
        for n in ("Ai_ez", "Ai_nz", "Ai_bx", "Ai_by"):
            a[n] = _f((imt, km, jmt, 2, 2))
        for n in ("adv_vbtiso", "adv_fb", "diff_fb", "diff_fbiso"):
            a[n] = _f((imt, km + 1, jmt))
        self.ctx = OrcCtx()
        self.ctx.imt, self.ctx.jmt, self.ctx.km, self.ctx.nt, self.ctx.nsrc = imt, jmt, km, nt, cfg.nsrc
        self.ctx.c2dtts = 2.0 * prm.dtts
        self.ctx.aidif = prm.aidif
        self.ctx.diff_cet, self.ctx.diff_cnt = prm.diff_cet, prm.diff_cnt
        self.ctx.slmxr, self.ctx.ahisop, self.ctx.athkdf = 1.0 / prm.slmx, prm.ahisop, prm.athkdf
        self.rebind()
        self.lib = lib()

    def rebind(self):
        for name, typ in OrcCtx._fields_:
            if typ in (_D, _I):
                arr = self.a.get(name)
                if arr is None:
                    setattr(self.ctx, name, typ())
                else:
                    assert arr.flags.f_contiguous or arr.ndim <= 1, name
                    setattr(self.ctx, name, arr.ctypes.data_as(typ))

    def set_src(self, src):
        self.a["src"] = np.asfortranarray(src)
        self.rebind()

    # entry points -------------------------------------------------------------
    def isopyc(self):
        self.lib.orc_isopyc(ctypes.byref(self.ctx))

    def add_k33(self):
        jmt = self.ctx.jmt
        self.a["diff_cbt"][...] = 0.0
        self.a["diff_cbt"][:, :, 1:jmt - 1] = self.ocean.diff_cbt_bg[:, :, 1:jmt - 1] + self.a["K33"][:, :, 1:jmt - 1]

    def adv_flux(self, n):
        self.lib.orc_adv_flux(ctypes.byref(self.ctx), ctypes.c_int(n))

    def diff_flux(self, n):
        self.lib.orc_diff_flux(ctypes.byref(self.ctx), ctypes.c_int(n))

    def isoflux(self, n):
        self.lib.orc_isoflux(ctypes.byref(self.ctx), ctypes.c_int(n))

    def explicit_update(self, n):
        self.lib.orc_explicit_update(ctypes.byref(self.ctx), ctypes.c_int(n))

    def transport(self):
        self.lib.orc_tracer_transport(ctypes.byref(self.ctx))
        return self.a["t_taup1"]


# ---- producers of the shared inputs (prep_oracle.c) ---------------------------------------------
def _p(a):
    return a.ctypes.data_as(_D)


def adv_vel(g, u):
    """adv_vet, adv_vnt (imt,km,jmt), adv_vbt (imt,km+1,jmt) from u(imt,km,jmt,2) as adv_vel.F."""
    imt, jmt, km = g.imt, g.jmt, g.km
    u1 = np.asfortranarray(u[..., 0]); u2 = np.asfortranarray(u[..., 1])
    vet, vnt, vbt = _f((imt, km, jmt)), _f((imt, km, jmt)), _f((imt, km + 1, jmt))
    c = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    arrs = [c(g.dxu), c(g.dyu), c(g.dxt2r), c(g.dyt2r), c(g.dxtr), c(g.dytr), c(g.cstr), c(g.csu), c(g.dzt)]
    lib().orc_adv_vel(ctypes.c_int(imt), ctypes.c_int(jmt), ctypes.c_int(km), _p(u1), _p(u2), *[_p(a) for a in arrs],
                      _p(vet), _p(vnt), _p(vbt))
    return vet, vnt, vbt


def vmixc(g, topo, tidal, alphai, betai, ddzt, K33, diff_cbt_prev):
    """diff_cbt (imt,km,jmt) as vmixc.F leaves it (tidal mixing + K33) from the isopyc products."""
    imt, jmt, km = g.imt, g.jmt, g.km
    out = np.array(diff_cbt_prev, order="F", dtype=np.float64)
    kmt = np.asfortranarray(topo.kmt, dtype=np.int32)
    c = lambda a: np.asfortranarray(a, dtype=np.float64)
    tl, zw = c(tidal.tlat), np.ascontiguousarray(g.zw, dtype=np.float64)
    e = [c(tidal.edrm2), c(tidal.edrs2), c(tidal.edrk1), c(tidal.edro1)]
    al, be, dz, k33 = c(alphai), c(betai), c(ddzt), c(K33)
    D = ctypes.c_double
    lib().orc_vmixc(ctypes.c_int(imt), ctypes.c_int(jmt), ctypes.c_int(km), kmt.ctypes.data_as(_I), _p(tl), _p(zw), _p(al),
                    _p(be), _p(dz), _p(k33), *[_p(a) for a in e], D(tidal.kappa_h), D(tidal.zetar), D(tidal.ogamma),
                    D(tidal.gravrho0r), _p(out))
    return out


# ---- polar Fourier filter (filter_oracle.c) ------------------------------------------------------
def findex(kmt, flt):
    """istf, ietf (jmtfil, lsegf, km) for the filtered rows as findex.F (O_cyclic)."""
    imt, jmt = kmt.shape
    km = flt.km
    kx = np.asfortranarray(kmt, dtype=np.int32)
    isf = np.zeros((flt.jmtfil, flt.lsegf, km), dtype=np.int32, order="F")
    ief = np.zeros_like(isf)
    rc = lib().orc_findex(kx.ctypes.data_as(_I), ctypes.c_int(imt), ctypes.c_int(jmt), ctypes.c_int(km), ctypes.c_int(flt.jfrst),
                          ctypes.c_int(flt.jft1), ctypes.c_int(flt.jft2), ctypes.c_int(flt.lsegf), ctypes.c_int(flt.jmtfil),
                          isf.ctypes.data_as(_I), ief.ctypes.data_as(_I))
    if rc:
        raise RuntimeError(f"orc_findex: more strips than lsegf or rows than jmtfil (rc={rc})")
    return isf, ief


def filt(t_taup1, g, topo, flt, istf, ietf, js=2, je=None):
    """Fourier-filter rows js..je of t(imt,km,jmt,nt) in place as filt.F does inside `tracer`."""
    imt, km, jmt, nt = t_taup1.shape
    je = jmt - 1 if je is None else je
    assert t_taup1.flags.f_contiguous
    kx = np.asfortranarray(topo.kmt, dtype=np.int32)
    cst, cstr = np.ascontiguousarray(g.cst), np.ascontiguousarray(g.cstr)
    rc = lib().orc_filt(_p(t_taup1), ctypes.c_int(imt), ctypes.c_int(km), ctypes.c_int(jmt), ctypes.c_int(nt),
                        kx.ctypes.data_as(_I), _p(cst), _p(cstr), ctypes.c_double(g.pi), ctypes.c_int(flt.jfrst),
                        ctypes.c_int(flt.jft0), ctypes.c_int(flt.jft1), ctypes.c_int(flt.jft2), ctypes.c_int(flt.lsegf),
                        ctypes.c_int(flt.jmtfil), istf.ctypes.data_as(_I), ietf.ctypes.data_as(_I), ctypes.c_int(js),
                        ctypes.c_int(je))
    if rc:
        raise RuntimeError("orc_filt: filtr hit one of the reference's stop conditions")
    return t_taup1


def findex_u(kmu, flt):
    """isuf, ieuf (jmtfil, lsegf, km): findex.F on kmu with the U rows (setmom.F:730)."""
    imt, jmt = kmu.shape
    km = flt.km
    kx = np.asfortranarray(kmu, dtype=np.int32)
    isf = np.zeros((flt.jmtfil, flt.lsegf, km), dtype=np.int32, order="F")
    ief = np.zeros_like(isf)
    rc = lib().orc_findex(kx.ctypes.data_as(_I), ctypes.c_int(imt), ctypes.c_int(jmt), ctypes.c_int(km), ctypes.c_int(flt.jfrst),
                          ctypes.c_int(flt.jfu1), ctypes.c_int(flt.jfu2), ctypes.c_int(flt.lsegf), ctypes.c_int(flt.jmtfil),
                          isf.ctypes.data_as(_I), ief.ctypes.data_as(_I))
    if rc:
        raise RuntimeError(f"orc_findex: more strips than lsegf or rows than jmtfil (rc={rc})")
    return isf, ief


def filuv(u_taup1, g, topo, mom, flt, js=2, je=None):
    """source/common/filuv.F on u(imt,km,jmt,2) = u(:,:,:,:,taup1) as called at clinic.F:500, followed by the
    setbcx of clinic.F:506-509; returns a new array."""
    imt, km, jmt, _ = u_taup1.shape
    je = jmt - 1 if je is None else je
    u1, u2 = np.array(u_taup1[..., 0], order="F"), np.array(u_taup1[..., 1], order="F")
    isuf, ieuf = findex_u(topo.kmu, flt)
    kx = np.asfortranarray(topo.kmu, dtype=np.int32)
    c = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    arrs = [c(g.csu), c(g.csur), c(g.phi), c(flt.spsin), c(flt.spcos), c(g.dzt), np.asfortranarray(mom.hr)]
    I = ctypes.c_int
    rc = lib().orc_filuv(_p(u1), _p(u2), I(imt), I(km), I(jmt), kx.ctypes.data_as(_I), *[_p(a) for a in arrs], ctypes.c_double(g.pi),
                         I(flt.jfrst), I(flt.jfu0), I(flt.jfu1), I(flt.jfu2), I(flt.lsegf), I(flt.jmtfil),
                         isuf.ctypes.data_as(_I), ieuf.ctypes.data_as(_I), I(js), I(je))
    if rc:
        raise RuntimeError("orc_filuv: filtr hit one of the reference's stop conditions")
    return np.stack([u1, u2], axis=-1)


def setbcx(a):
    """Cyclic images of columns 2 and imt-1 (source/common/util.F:789-814), in place on the first axis."""
    a[0] = a[-2]
    a[-1] = a[1]
    return a


# ---- baroclinic momentum step (clinic_oracle.c) ------------------------------
_MOM_SCAL = ["c2dtuv", "grav", "rho0r", "kappa_m", "cdbot"]
_MOM_1D = ["dxur", "dxu2r", "dxtr", "dxmetr", "duw", "due", "dyur", "dyu2r", "dyu4r", "dytr", "csur", "cst", "dus", "dun",
           "csudyu2r", "advmet", "am3", "am4", "dzt", "dztr", "dzt2r", "dzw", "dzwr"]
_MOM_A = ["umask", "hr", "cori", "visc_ceu", "amc_north", "amc_south", "adv_vet", "adv_vnt", "adv_vbt", "smf", "rho"]
_MOM_OUT = ["zu", "bmf", "adv_veu", "adv_vnu", "adv_vbu", "grad_p"]


class OrcMom(ctypes.Structure):
    _fields_ = ([(n, ctypes.c_int) for n in ("imt", "jmt", "km")] + [(n, ctypes.c_double) for n in _MOM_SCAL]
                + [(n, _D) for n in _MOM_1D] + [("kmu", _I)] + [(n, _D) for n in _MOM_A]
                + [("u_tau", _D * 2), ("u_taum1", _D * 2), ("u_taup1", _D * 2)] + [(n, _D) for n in _MOM_OUT])


def state(g, eos, t, s, js=2, je=None):
    """rho (imt,km,jmt) from T,S (imt,km,jmt) as source/mom/state.F, rows js..je (loadmw.F:154: 2..jmt)."""
    imt, jmt, km = g.imt, g.jmt, g.km
    je = jmt if je is None else je
    rho = _f((imt, km, jmt))
    c = lambda a: np.asfortranarray(a, dtype=np.float64)
    tt, ss, to, so, cc = c(t), c(s), c(eos[0]), c(eos[1]), c(eos[2])      # eos = synthetic.load_eos(km)
    lib().orc_state(ctypes.c_int(imt), ctypes.c_int(jmt), ctypes.c_int(km), _p(tt), _p(ss), _p(to), _p(so), _p(cc), _p(rho),
                    ctypes.c_int(js), ctypes.c_int(je))
    return rho


class Momentum:
    """One baroclinic momentum step of the oracle on a synthetic ocean: m = Momentum(ocean, mom, rho);
    m.adv_vel_u(); m.bmf(); m.clinic() -> arrays in m.a (u_taup1 as (imt,km,jmt,2), zu, grad_p, ...)."""

    def __init__(self, ocean, mom, rho, u_tau=None, u_taum1=None):
        g, topo = ocean.grid, ocean.topo
        imt, jmt, km = g.imt, g.jmt, g.km
        self.g = g
        c = lambda a: np.asfortranarray(a, dtype=np.float64)
        a = self.a = {}
        ctx = self.ctx = OrcMom()
        ctx.imt, ctx.jmt, ctx.km = imt, jmt, km
        ctx.c2dtuv = 2.0 * mom.dtuv
        ctx.grav, ctx.rho0r, ctx.kappa_m, ctx.cdbot = mom.grav, mom.rho0r, mom.kappa_m, mom.cdbot
        for n in _MOM_1D:
            a[n] = c(getattr(mom, n) if hasattr(mom, n) else getattr(g, n))
            setattr(ctx, n, _p(a[n]))
        a["kmu"] = np.asfortranarray(topo.kmu, dtype=np.int32)
        ctx.kmu = a["kmu"].ctypes.data_as(_I)
        src = {"umask": topo.umask, "hr": mom.hr, "cori": mom.cori, "visc_ceu": mom.visc_ceu, "amc_north": mom.amc_north,
               "amc_south": mom.amc_south, "adv_vet": ocean.adv_vet, "adv_vnt": ocean.adv_vnt, "adv_vbt": ocean.adv_vbt,
               "smf": mom.smf, "rho": rho}
        for n in _MOM_A:
            a[n] = c(src[n])
            setattr(ctx, n, _p(a[n]))
        ut = ocean.u if u_tau is None else u_tau
        um = mom.u_taum1 if u_taum1 is None else u_taum1
        a["u_tau"] = [c(ut[..., 0]), c(ut[..., 1])]
        a["u_taum1"] = [c(um[..., 0]), c(um[..., 1])]
        a["u_taup1"] = [_f((imt, km, jmt)), _f((imt, km, jmt))]
        for n in ("u_tau", "u_taum1", "u_taup1"):
            setattr(ctx, n, (_D * 2)(_p(a[n][0]), _p(a[n][1])))
        shapes = {"zu": (imt, jmt, 2), "bmf": (imt, jmt, 2), "adv_veu": (imt, km, jmt), "adv_vnu": (imt, km, jmt),
                  "adv_vbu": (imt, km + 1, jmt), "grad_p": (imt, km, jmt, 2)}
        for n in _MOM_OUT:
            a[n] = _f(shapes[n])
            setattr(ctx, n, _p(a[n]))
        self.kmt = np.asfortranarray(topo.kmt, dtype=np.int32)

    def adv_vel_u(self):
        lib().orc_adv_vel_u(ctypes.byref(self.ctx))
        return self.a["adv_veu"], self.a["adv_vnu"], self.a["adv_vbu"]

    def bmf(self):
        lib().orc_bmf(ctypes.byref(self.ctx))
        return self.a["bmf"]

    def clinic(self):
        lib().orc_clinic(ctypes.byref(self.ctx))
        return np.stack(self.a["u_taup1"], axis=-1), self.a["zu"]

    def step(self):
        self.adv_vel_u()
        self.bmf()
        return self.clinic()

    def add_ext_mode(self, psi, u):
        """loadmw.F's add_ext_mode for one time level: psi (imt,jmt), u (imt,km,jmt,2); returns a new array."""
        u1, u2 = np.array(u[..., 0], order="F"), np.array(u[..., 1], order="F")
        ps = np.asfortranarray(psi, dtype=np.float64)
        lib().orc_add_ext_mode(ctypes.byref(self.ctx), _p(ps), _p(u1), _p(u2))
        return np.stack([u1, u2], axis=-1)

    def sbcu(self, which, sbc_u, sbc_v, osegs, osege, rts):
        """isbcu ("i") / asbcu ("a") in place on two (imt,jmt) planes."""
        f = lib().orc_isbcu if which == "i" else lib().orc_asbcu
        f(ctypes.byref(self.ctx), _p(sbc_u), _p(sbc_v), ctypes.c_int(int(osegs)), ctypes.c_int(int(osege)), ctypes.c_double(rts),
          self.kmt.ctypes.data_as(_I))
