"""Drive the compiled reference (oracle/_ref) on a synthetic ocean.

TEST INFRASTRUCTURE ONLY.  Fills the reference's COMMON blocks from an
`uvic29_amd.synthetic.Ocean`, runs the reference's own initialisation routines
that do not need netCDF (`tracer_init`, `eqstate`, `isopi`, `mobi_init`) and
then calls the reference's `isopyc` / `tracer` (or any single routine of the
path) exactly as `mom` does (/root/reference/source/mom/mom.F:325-405).
Call order follows SURVEY.md Appendix B.
"""
from __future__ import annotations

import os
import re
import tempfile
from pathlib import Path

import numpy as np

from refmodel import RefLib

REF = Path(os.environ.get("UVIC_REFERENCE", "/root/reference"))

SILICON_ONLY = ["abiodiat", "kfemin_Diat", "kfemax_Diat", "nu_diat", "nudt0", "wo0", "sipr0",
                "sildustfluxfac", "zprefDiat"]


def trimmed_control_in(cfg_options) -> str:
    """run/control.in with the silicon-only &mobi members removed for option
    sets without O_mobi_silicon (they are namelist members only under that
    option, mobi.F:41-44,53-57)."""
    text = (REF / "run" / "control.in").read_text()
    if "mobi_silicon" in cfg_options:
        return text
    for name in SILICON_ONLY:
        text = re.sub(r"\b" + name + r"\s*=\s*[-+0-9.eE]+\s*,?", "", text)
    return text


class RefOcean:
    """The reference model state initialised from a synthetic Ocean."""

    def __init__(self, ocean, quiet: bool = True, shim: bool = False):
        g, cfg = ocean.grid, ocean.cfg
        self.ocean = ocean
        self.ref = RefLib(cfg.name, g.imt, g.jmt, g.km, shim=shim)
        self.v = self.ref.v
        self.quiet = quiet
        if shim and hasattr(self.ref.lib, "tracer_gpu_close_"):
            # the library is loaded once per process and the overlays keep their device instance and what they know about
            # it in module variables: a new model instance starts from nothing (environment switches are read again)
            self.ref.call("tracer_gpu_close")
        self._init(ocean)

    # -- helpers ---------------------------------------------------------------
    def _silence(self):
        """Redirect the Fortran runtime's stdout (fd 1) while init prints."""
        if not self.quiet:
            return None
        import sys
        sys.stdout.flush()
        saved = os.dup(1)
        devnull = os.open(os.devnull, os.O_WRONLY)
        os.dup2(devnull, 1)
        os.close(devnull)
        return saved

    def _restore(self, saved):
        if saved is None:
            return
        self.ref.lib.orc_flush_()      # drain the Fortran runtime's unit-6 buffer into /dev/null
        os.dup2(saved, 1)
        os.close(saved)

    def setv(self, name, value):
        self.v[name][...] = value

    # -- initialisation --------------------------------------------------------
    def _init(self, oc):
        g, topo, cfg, prm, v = oc.grid, oc.topo, oc.cfg, oc.params, self.v
        imt, jmt, km = g.imt, g.jmt, g.km
        S = self.ref.set
        S("pi", g.pi); S("radian", g.radian); S("radius", 6370.0e5)
        S("grav", 980.6); S("rho0", 1.035); S("rho0r", 1.0 / 1.035)
        S("daylen", 86400.0)
        S("dtts", prm.dtts); S("c2dtts", 2.0 * prm.dtts)
        S("aidif", prm.aidif); S("kappa_h", prm.kappa_h)
        S("diff_cet", prm.diff_cet); S("diff_cnt", prm.diff_cnt); S("ah", prm.diff_cet)
        S("taum1", -1); S("tau", 0); S("taup1", 1)
        S("eots", 1); S("first", 1); S("euler2", 0)
        # one library = one set of COMMON blocks per process: a new instance starts from plain leapfrog steps with every
        # diagnostic switch off, whatever the instance before it left behind
        S("euler1", 0); S("forward", 0); S("leapfrog", 1)
        for n in ("tavgts", "trmbts", "gyrets", "glents", "xbtperts", "timavgperts", "tsiperts", "stabts", "cmixts", "osegs", "osege"):
            S(n, 0)
        S("relyr", oc.forcing.relyr)
        S("jfrst", jmt + 1)            # polar filter off
        for n in ("xt", "yt", "xu", "yu", "zw", "zt", "dxtdeg", "dytdeg", "dzt", "dxudeg", "dyudeg", "dzw",
                  "dxt", "dxtr", "dxt2r", "dxu", "dxur", "dxu2r", "dxu4r", "dxt4r",
                  "dyt", "dytr", "dyt2r", "dyu", "dyur", "dyu2r", "dyu4r", "dyt4r",
                  "csu", "csur", "cst", "cstr", "cstdytr", "cstdyt2r", "csudyur", "csudyu2r",
                  "cst_dytr", "csu_dyur", "phi", "phit", "sine", "tng", "c2dzt", "dztr", "dzt2r",
                  "dzwr", "dzw2r", "dztur", "dztlr", "tlat", "tlon",
                  "dtxcel", "dtxsqr", "dztxcl", "dzwxcl"):
            v[n][...] = getattr(g, n)
        v["kmt"][...] = topo.kmt
        v["kmu"][...] = topo.kmu
        v["sg_bathy"][...] = topo.sg_bathy
        v["tmask"][...] = topo.tmask
        v["umask"][...] = topo.umask
        saved = self._silence()
        try:
            self.ref.call("tracer_init")
            self.ref.call("eqstate", v["zt"], km, v["ro0"], v["to"], v["so"], v["c"],
                          v["tmink"], v["tmaxk"], v["smink"], v["smaxk"])
            cwd = os.getcwd()
            with tempfile.TemporaryDirectory() as td:
                os.chdir(td)
                try:
                    err = np.zeros(1, dtype=np.int32)
                    self.ref.call("isopi", err, 1.5e9, prm.diff_cet)
                    if cfg.ntnpzd and (REF / "run" / "control.in").exists():
                        Path("control.in").write_text(trimmed_control_in(cfg.options))
                        self.ref.call("mobi_init")
                    elif cfg.ntnpzd:
                        self._mobi_from_fixture(cfg, km)
                finally:
                    os.chdir(cwd)
        finally:
            self._restore(saved)
        # isopi hard-codes slmx/ahisop/athkdf (isopyc.F:70-86); overwrite with the ocean's
        S("slmxr", 1.0 / prm.slmx); S("ahisop", prm.ahisop); S("athkdf", prm.athkdf)
        v["fisop"][...] = oc.fisop
        v["addisop"][...] = oc.addisop[:, :, 1:jmt - 1]
        # state
        self.load_state(oc.t_taum1, oc.t_tau)
        v["u"][:, :, :, :, 1] = oc.u           # tau slot (index -1:1 -> 0,1,2)
        v["adv_vet"][...] = oc.adv_vet[:, :, 1:]
        v["adv_vnt"][...] = oc.adv_vnt
        v["adv_vbt"][...] = oc.adv_vbt[:, :, 1:]
        v["stf"][...] = oc.stf
        v["btf"][...] = oc.btf
        if cfg.ntnpzd:
            f = oc.forcing
            v["dnswr"][...] = f.dnswr
            v["aice"][:, :, 1] = f.aice
            v["hice"][:, :, 1] = f.hice
            v["hsno"][:, :, 1] = f.hsno
            S("co2ccn", f.co2ccn)
            v["fe_atmdep"][:, :, 0, :] = f.fe_atmdep
            v["fe_hydr"][...] = f.fe_hydr
        self.diff_cbt_bg = oc.diff_cbt_bg

    def _mobi_from_fixture(self, cfg, km):
        """GPU box: /root/reference (hence run/control.in) is absent, so COMMON /npzd_r/
        and the imobi* indices are set from the committed fixture that mobi_init produced
        in the build container (uvic2.9_amd/data/mobi_<cfg>.json)."""
        from uvic29_amd import mobi as pm
        tab = pm.load_table(cfg.name, km)
        for n, val in tab.items():
            if n == "imobi":      # the column order: tracer_init has set it; recorded in the table for the tests only
                continue
            self.v[n][...] = np.asarray(val).reshape(self.v[n].shape, order="F")
        for m, name in enumerate(cfg.mobi):
            self.ref.set("imobi" + name, m + 1)

    def load_state(self, t_taum1, t_tau):
        self.v["t"][..., 0] = t_taum1
        self.v["t"][..., 1] = t_tau
        self.v["t"][..., 2] = 0.0

    # -- one ocean tracer step as mom.F does -----------------------------------
    def isopyc(self):
        g = self.ocean.grid
        self.ref.call("isopyc", 0, 1, g.jmt, 2, g.imt - 1)

    def add_k33(self):
        """diff_cbt = background + K33 (updates/09/source/mom/vmixc.F:182-188)."""
        g = self.ocean.grid
        self.v["diff_cbt"][...] = self.diff_cbt_bg[:, :, 1:g.jmt - 1] + self.v["k33"]

    def tracer(self):
        g = self.ocean.grid
        self.ref.call("tracer", 0, 2, g.jmt - 1, 2, g.imt - 1)

    # -- polar Fourier filter -------------------------------
    def set_filter(self, flt):
        """Switch the polar filter of `tracer` on: filter rows as setcom.F computes them, strips by the
        reference's own findex (setmom.F:729)."""
        S, v = self.ref.set, self.v
        if flt.jmtfil != v["istf"].shape[0] or flt.lsegf != v["istf"].shape[1]:
            raise ValueError("filter dimensions differ from the reference build (index.h: jmtfil=50, lsegf=20)")
        S("jfrst", flt.jfrst); S("jft0", flt.jft0); S("jft1", flt.jft1); S("jft2", flt.jft2); S("jskpt", flt.jskpt)
        S("njtbft", flt.njtbft)
        v["istf"][...] = 0
        v["ietf"][...] = 0
        g = self.ocean.grid
        self.ref.call("findex", v["kmt"], flt.jmtfil, g.km, flt.jft1, flt.jft2, g.imt, v["istf"], v["ietf"])
        return np.array(v["istf"], order="F"), np.array(v["ietf"], order="F")

    def filt(self):
        """source/common/filt.F on t(taup1) as called at tracer.F:1245."""
        g = self.ocean.grid
        saved = self._silence()
        try:
            self.ref.call("filt", 0, 2, g.jmt - 1)
        finally:
            self._restore(saved)
        return self.v["t"][..., 2]

    # -- producers of the shared inputs -------------------
    def adv_vel(self):
        """source/mom/adv_vel.F with the arguments of mom.F:332 for one memory window; returns
        adv_vet, adv_vnt, adv_vbt over all jmt rows (row 1 of vet/vbt is not computed: zero)."""
        g = self.ocean.grid
        for n in ("adv_vet", "adv_vnt", "adv_vbt"):
            self.v[n][...] = 0.0
        self.ref.call("adv_vel", 0, 1, g.jmt, 2, g.imt - 1)
        vet = np.zeros((g.imt, g.km, g.jmt), order="F"); vet[:, :, 1:] = self.v["adv_vet"]
        vbt = np.zeros((g.imt, g.km + 1, g.jmt), order="F"); vbt[:, :, 1:] = self.v["adv_vbt"]
        return vet, np.array(self.v["adv_vnt"], order="F"), vbt

    def vmixc(self, tidal, diff_cbt_prev):
        """updates/09/source/mom/vmixc.F (mom.F:347) after `isopyc`: tidal mixing + K33.  Returns
        diff_cbt over all jmt rows (rows 1 and jmt are not part of the reference array: zero)."""
        g = self.ocean.grid
        S = self.ref.set
        for n in ("edrm2", "edrs2", "edrk1", "edro1"):
            self.v[n][...] = getattr(tidal, n)
        S("zetar", tidal.zetar); S("ogamma", tidal.ogamma); S("gravrho0r", tidal.gravrho0r); S("kappa_h", tidal.kappa_h)
        self.v["diff_cbt"][...] = diff_cbt_prev[:, :, 1:g.jmt - 1]
        self.ref.call("vmixc", 0, 1, g.jmt, 2, g.imt - 1)
        out = np.zeros((g.imt, g.km, g.jmt), order="F")
        out[:, :, 1:g.jmt - 1] = self.v["diff_cbt"]
        return out

    # -- baroclinic momentum step (configuration "m2" of build_ref.py) ----------
    def set_momentum(self, mom):
        """COMMON data of `clinic` that the tracer path does not use: u(tau-1), wind stress (through sbc, as
        setvbc reads it), the factors setmom.F computes once and the coefficients hmixc.F leaves on its first call."""
        g, v, S = self.ocean.grid, self.v, self.ref.set
        for n in ("dxmetr", "duw", "due", "dus", "dun"):
            v[n][...] = getattr(g, n)
        for n in ("cori", "am3", "am4", "advmet", "hr"):
            v[n][...] = getattr(mom, n)
        if v["amc_north"].ndim == 3:          # O_anisotropic_viscosity: three-dimensional coefficients (hmixc.h)
            for n in ("visc_ceu", "amc_north", "amc_south"):
                v[n][...] = getattr(mom, n)
        else:                                 # one per row; visc_ceu = visc_cnu = am
            S("visc_ceu", mom.am); S("visc_cnu", mom.am)
            v["amc_north"][...] = mom.amc_north_row
            v["amc_south"][...] = mom.amc_south_row
        S("am", mom.am); S("kappa_m", mom.kappa_m); S("cdbot", mom.cdbot)
        S("dtuv", mom.dtuv); S("c2dtuv", 2.0 * mom.dtuv); S("acor", 0.0)
        v["visc_cbu"][...] = mom.kappa_m        # u09/mom/vmixc.F:85 (O_constvmix)
        v["u"][:, :, :, :, 0] = mom.u_taum1
        if int(self.ref.get("itaux")) == 0:      # no coupler initialisation here: give the stresses two free sbc planes
            S("itaux", v["sbc"].shape[2] - 1); S("itauy", v["sbc"].shape[2])
        v["sbc"][:, :, int(self.ref.get("itaux")) - 1] = mom.smf[..., 0]
        v["sbc"][:, :, int(self.ref.get("itauy")) - 1] = mom.smf[..., 1]
        self.mom = mom

    def set_filter_u(self, flt):
        """Switch `filuv` inside clinic on: U rows as setcom.F computes them, strips by the reference's findex on kmu
        (setmom.F:730), rotation factors of setcom.F:55-70."""
        S, v = self.ref.set, self.v
        if flt.jmtfil != v["isuf"].shape[0] or flt.lsegf != v["isuf"].shape[1]:
            raise ValueError("filter dimensions differ from the reference build (index.h: jmtfil=50, lsegf=20)")
        S("jfrst", flt.jfrst); S("jfu0", flt.jfu0); S("jfu1", flt.jfu1); S("jfu2", flt.jfu2); S("jskpu", flt.jskpu)
        S("njtbfu", flt.njtbfu)
        v["spsin"][...] = flt.spsin
        v["spcos"][...] = flt.spcos
        v["isuf"][...] = 0
        v["ieuf"][...] = 0
        g = self.ocean.grid
        self.ref.call("findex", v["kmu"], flt.jmtfil, g.km, flt.jfu1, flt.jfu2, g.imt, v["isuf"], v["ieuf"])
        return np.array(v["isuf"], order="F"), np.array(v["ieuf"], order="F")

    def add_ext_mode(self, psi, level="tau"):
        """loadmw.F's add_ext_mode on u(tau) / u(tau-1) with psi (imt,jmt) as the stream function of that level."""
        g, v = self.ocean.grid, self.v
        v["psi"][:, :, 0 if level == "tau" else 1] = psi
        lev = level.encode()
        import ctypes
        # character(*) dummy: the length follows the arguments, by value (flang / gfortran convention)
        fn = getattr(self.ref.lib, "add_ext_mode_")
        I = lambda x: ctypes.byref(ctypes.c_int(x))
        fn(I(0), I(1), I(g.jmt), I(2), I(g.imt - 1), ctypes.c_char_p(lev), ctypes.c_size_t(len(lev)))
        return np.array(v["u"][..., 1 if level == "tau" else 0], order="F")

    def state(self):
        """rho as loadmw.F:154 computes it: rows 2..jmt, all columns; returned over all jmt rows (row 1 zero)."""
        g, v = self.ocean.grid, self.v
        t = v["t"]
        T = np.array(t[:, :, :, 0, 1], order="F"); Sa = np.array(t[:, :, :, 1, 1], order="F")
        self.ref.call("state", T, Sa, v["rho"], 2, g.jmt, 1, g.imt)
        rho = np.zeros((g.imt, g.km, g.jmt), order="F"); rho[:, :, 1:] = v["rho"]
        return rho

    def adv_vel_u(self):
        """adv_vel.F (all of it) as called at mom.F:332; returns adv_veu, adv_vnu, adv_vbu over all jmt rows."""
        g, v = self.ocean.grid, self.v
        self.ref.call("adv_vel", 0, 1, g.jmt, 2, g.imt - 1)
        veu = np.zeros((g.imt, g.km, g.jmt), order="F"); veu[:, :, 1:g.jmt - 1] = v["adv_veu"]
        vnu = np.zeros((g.imt, g.km, g.jmt), order="F"); vnu[:, :, :g.jmt - 1] = v["adv_vnu"]
        vbu = np.zeros((g.imt, g.km + 1, g.jmt), order="F"); vbu[:, :, 1:g.jmt - 1] = v["adv_vbu"]
        return veu, vnu, vbu

    def setvbc(self):
        """u09/mom/setvbc.F as called at mom.F:375: smf from sbc, bmf from u(tau-1); both (imt,jmt,2)."""
        g = self.ocean.grid
        self.ref.call("setvbc", 0, 2, g.jmt - 1, 2, g.imt - 1)
        return np.array(self.v["smf"], order="F"), np.array(self.v["bmf"], order="F")

    def clinic(self):
        """u09/mom/clinic.F as called at mom.F:395.  Returns u(tau+1) (imt,km,jmt,2), zu (imt,jmt,2) and
        grad_p (imt,km,jmt,2; rows 1 and jmt zero)."""
        g, v = self.ocean.grid, self.v
        saved = self._silence()
        try:
            self.ref.call("clinic", 0, 2, g.jmt - 1, 2, g.imt - 1)
        finally:
            self._restore(saved)
        gp = np.zeros((g.imt, g.km, g.jmt, 2), order="F"); gp[:, :, 1:g.jmt - 1] = v["grad_p"]
        return np.array(v["u"][..., 2], order="F"), np.array(v["zu"], order="F"), gp

    def set_tidal(self, tidal):
        """COMMON /tdr/ (updates/09/source/mom/tidal_kv.h) and the diffusivity vmixc finds below the bottom level on
        the first step: from here on `step` calls the reference's own vmixc (tidal mixing + K33) after isopyc."""
        g = self.ocean.grid
        for n in ("edrm2", "edrs2", "edrk1", "edro1"):
            self.v[n][...] = getattr(tidal, n)
        for n in ("zetar", "ogamma", "gravrho0r", "kappa_h"):
            self.ref.set(n, getattr(tidal, n))
        self.v["diff_cbt"][...] = self.diff_cbt_bg[:, :, 1:g.jmt - 1]
        self.v["k33"][...] = 0.0
        self.tidal = tidal

    def step(self, c2dtts=None):
        if c2dtts is not None:
            self.ref.set("c2dtts", c2dtts)
        self.isopyc()
        if getattr(self, "tidal", None) is not None:
            g = self.ocean.grid
            self.ref.call("vmixc", 0, 1, g.jmt, 2, g.imt - 1)       # mom.F:347
        else:
            self.add_k33()
        self.tracer()
        return self.v["t"][..., 2]

    def rotate(self):
        """taum1 <- tau, tau <- taup1 (what putmw/getvar do through the ramdrive)."""
        t = self.v["t"]
        t[..., 0] = t[..., 1]
        t[..., 1] = t[..., 2]

    def set_step_kind(self, forward: bool):
        """A forward ("mixing") step: c2dtts = dtts and both MW slots hold the tau data
        (updates/09/source/mom/loadmw.F:107-111); else leapfrog with c2dtts = 2 dtts."""
        dtts = self.ocean.params.dtts
        self.ref.set("forward", 1 if forward else 0)
        self.ref.set("leapfrog", 0 if forward else 1)
        self.ref.set("c2dtts", dtts if forward else 2.0 * dtts)
        if forward:
            self.v["t"][..., 0] = self.v["t"][..., 1]

    def flush(self):
        """Resident overlay only: the device's t(tau-1), t(tau) into the host's slots."""
        self.ref.call("tracer_gpu_flush")


# ---- the reference's second boundary: `tracer` built with -DO_TMM -----------------------------
def tmm_reference_sources(ocean, cols, c2dtts=None, quiet=True):
    """Source terms of a batch of columns from the reference compiled with O_TMM (build "tmm30": imt = batch size,
    jmt = 1; `tracer` is then only the MOBI source loop + the 14C source and publishes src in COMMON /mobicomm/,
    updates/09/source/mom/tracer.F:109-124).  `cols`: list of 1-based (i, j) of the synthetic ocean; returns
    src (ncols, km, nsrc)."""
    import ctypes
    g, cfg, prm, f, topo = ocean.grid, ocean.cfg, ocean.params, ocean.forcing, ocean.topo
    ncols, km = len(cols), g.km
    ref = RefLib("tmm30", ncols, 1, km)
    v, S = ref.v, ref.set
    ii = np.array([c[0] - 1 for c in cols]); jj = np.array([c[1] - 1 for c in cols])
    S("pi", g.pi); S("radian", g.radian); S("daylen", 86400.0)
    S("dtts", prm.dtts); S("c2dtts", 2.0 * prm.dtts if c2dtts is None else c2dtts)
    S("taum1", -1); S("tau", 0); S("taup1", 1)
    S("eots", 1); S("relyr", f.relyr)
    for n in ("zw", "zt", "dzt", "dzw", "dztr", "dzt2r", "dzwr"):
        v[n][...] = getattr(g, n)
    v["kmt"][:, 0] = topo.kmt[ii, jj]
    v["sg_bathy"][:, 0, :] = topo.sg_bathy[ii, jj, :]
    v["tlat"][:, 0] = g.tlat[ii, jj]
    saved = None
    if quiet:
        import sys
        sys.stdout.flush()
        saved = os.dup(1)
        devnull = os.open(os.devnull, os.O_WRONLY); os.dup2(devnull, 1); os.close(devnull)
    try:
        ref.call("tracer_init")
        cwd = os.getcwd()
        with tempfile.TemporaryDirectory() as td:
            os.chdir(td)
            try:
                Path("control.in").write_text(trimmed_control_in(cfg.options))
                ref.call("mobi_init")
            finally:
                os.chdir(cwd)
    finally:
        if saved is not None:
            ref.lib.orc_flush_()
            os.dup2(saved, 1); os.close(saved)
    v["t"][:, :, 0, :, 0] = ocean.t_taum1[ii, :, jj, :]
    v["dnswr"][:, 0] = f.dnswr[ii, jj]
    v["aice"][:, 0, 1] = f.aice[ii, jj]
    v["hice"][:, 0, 1] = f.hice[ii, jj]
    v["hsno"][:, 0, 1] = f.hsno[ii, jj]
    S("co2ccn", f.co2ccn)
    v["fe_atmdep"][:, 0, 0, :] = f.fe_atmdep[ii, jj, :]
    v["fe_hydr"][:, 0, :] = f.fe_hydr[ii, jj, :]
    ref.call("tracer", 0, 1, 1, 1, ncols)
    n = ncols * km * cfg.nsrc
    buf = (ctypes.c_double * n).in_dll(ref.lib, "mobicomm_")
    return np.frombuffer(buf, dtype=np.float64, count=n).reshape((ncols, km, cfg.nsrc), order="F").copy()
