"""Host-side mirror of the reference's ocean tracer entry points.

`TracerModel` owns one device-resident model instance (C ABI handle) and exposes
the calls that `subroutine mom` makes on the tracer path, with the same names,
argument meaning and error behaviour:

    isopyc(joff, js, je, is_, ie)   <- /root/reference/source/mom/mom.F:340
    tracer(joff, js, je, is_, ie)   <- /root/reference/source/mom/mom.F:389

In the reference all bulk data travel through COMMON blocks; here the fields of
those blocks are device buffers addressed by name (`upload('t_taum1', a)`,
`download('t_taup1')`).  Arrays are numpy, Fortran order, shaped as listed in
include/uvic_gpu.h.  Everything computes on the GPU through libuvic_gpu.so.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

from . import capi
from .capi import FIELD, Dims, Params, UvicGpuError, check

_GRID_1D = ("dxt", "dxtr", "dxu", "dxur", "dxt4r", "dyt", "dytr", "dyu", "dyur", "dyt4r", "cst", "cstr", "csu",
            "cstdytr", "cstdyt2r", "csu_dyur", "dzt", "dztr", "dzt2r", "dztur", "dztlr", "dzw", "dzwr",
            "dtxcel", "dtxsqr", "dztxcl")


class TracerModel:
    def __init__(self, imt, jmt, km, nt, nsrc=0, ntnpzd=0, device=0):
        self.lib = capi.load()
        self.dims = Dims(imt, jmt, km, nt, nsrc, ntnpzd)
        self.imt, self.jmt, self.km, self.nt, self.nsrc = imt, jmt, km, nt, nsrc
        self.h = ctypes.c_void_p()
        check(self.lib.uvic_gpu_create(ctypes.byref(self.h), ctypes.byref(self.dims), device), "uvic_gpu_create")
        self.params = Params()
        self.device = device
        self.has_mobi = False

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.lib.uvic_gpu_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- data movement -----------------------------------------------------------
    def shape(self, name):
        imt, jmt, km, nt, nsrc = self.imt, self.jmt, self.km, self.nt, max(self.nsrc, 1)
        C, Fc, S = (imt, km, jmt), (imt, km + 1, jmt), (imt, jmt)
        table = {
            "kmt": S, "fisop": (imt, jmt, km), "addisop": C, "t_taum1": C + (nt,), "t_tau": C + (nt,),
            "t_taup1": C + (nt,), "adv_vet": C, "adv_vnt": C, "adv_vbt": Fc, "diff_cbt_bg": C,
            "stf": S + (nt,), "btf": S + (nt,), "src": C + (nsrc,), "itrc": (nt,), "alphai": C, "betai": C,
            "ddxt": C + (2,), "ddyt": C + (2,), "ddzt": Fc + (2,), "ai_ez": C + (2, 2), "ai_nz": C + (2, 2),
            "ai_bx": C + (2, 2), "ai_by": C + (2, 2), "k11": C, "k22": C, "k33": C, "adv_vetiso": C,
            "adv_vntiso": C, "adv_vbtiso": Fc, "diff_cbt": C, "to": (km,), "so": (km,), "c": (km, 9),
            "u1": C, "u2": C, "tlat": S, "edrm2": C, "edrs2": C, "edrk1": C, "edro1": C, "dxt2r": (imt,), "zw": (km,),
            "rho": C, "um1": C, "um2": C, "up1": C, "up2": C, "zu": S + (2,), "grad_p": C + (2,), "smf": S + (2,), "kmu": S,
            "hr": S, "cori": S + (2,), "visc_ceu": C, "amc_north": C, "amc_south": C, "dxu2r": (imt,), "dxmetr": (imt,),
            "duw": (imt,), "due": (imt,), "advmet": (jmt, 2), "am4": (jmt, 2), "sbc_gu": S, "sbc_gv": S, "sbc_su": S,
            "sbc_sv": S, "spsin": (imt,), "spcos": (imt,), "psi": S + (2,),
        }
        if name in table:
            return table[name]
        if name in ("dxt", "dxtr", "dxu", "dxur", "dxt4r"):
            return (imt,)
        if name in ("dzw", "dzwr"):
            return (km + 1,)
        if name in ("dzt", "dztr", "dzt2r", "dztur", "dztlr", "dtxcel", "dtxsqr", "dztxcl"):
            return (km,)
        return (jmt,)

    def upload(self, name, array):
        f = FIELD[name.lower()]
        dt = np.int32 if name.lower() in ("kmt", "itrc", "kmu") else np.float64
        a = np.asfortranarray(array, dtype=dt)
        if a.shape != self.shape(name.lower()):
            raise UvicGpuError(f"upload({name}): shape {a.shape} != {self.shape(name.lower())}")
        check(self.lib.uvic_gpu_upload(self.h, f, a.ctypes.data_as(ctypes.c_void_p), 0, a.size), f"upload({name})")

    def download(self, name):
        f = FIELD[name.lower()]
        dt = np.int32 if name.lower() in ("kmt", "itrc", "kmu") else np.float64
        a = np.zeros(self.shape(name.lower()), dtype=dt, order="F")
        check(self.lib.uvic_gpu_download(self.h, f, a.ctypes.data_as(ctypes.c_void_p), 0, a.size), f"download({name})")
        return a

    def upload_rows(self, name, array, jlo, jhi):
        """`array` dimensioned (imt, kdim, jlo:jhi[, extra]) as in the reference's COMMON."""
        a = np.asfortranarray(array, dtype=np.float64)
        check(self.lib.uvic_gpu_upload_rows(self.h, FIELD[name.lower()], a.ctypes.data_as(ctypes.c_void_p), jlo, jhi),
              f"upload_rows({name})")

    def devptr(self, name):
        return self.lib.uvic_gpu_field_devptr(self.h, FIELD[name.lower()])

    def set_params(self, **kw):
        for k, v in kw.items():
            setattr(self.params, k, v)
        check(self.lib.uvic_gpu_set_params(self.h, ctypes.byref(self.params)), "set_params")

    def set_mobi(self, ocean, table=None, generic=False):
        """Upload MOBI parameters (COMMON /npzd_r/ after mobi_init) and forcing; from
        then on `tracer` computes the source terms on the device.  generic: option set C through the general
        column kernel as well (cross-check)."""
        from . import mobi as pm
        prm = table if table is not None else pm.load_table(ocean.cfg.name, ocean.grid.km)
        P = pm.make_params(ocean.cfg, ocean.grid, prm)
        F = pm.Forcing(ocean)
        if generic:
            self.set_option("mobi_generic", 1)
        if pm.is_set_c(ocean.cfg) and not generic:
            check(self.lib.uvic_gpu_set_mobi(self.h, ctypes.byref(P), ctypes.byref(F.c)), "set_mobi")
        else:   # another option set of SURVEY.md §2c: flags and the extra parameters travel in uvic_mobi_options
            O = pm.make_options(ocean.cfg, ocean.grid, prm)
            check(self.lib.uvic_gpu_set_mobi_opt(self.h, ctypes.byref(P), ctypes.byref(O), ctypes.byref(F.c)), "set_mobi_opt")
        self.has_mobi = True

    def mobi(self):
        check(self.lib.uvic_gpu_mobi(self.h), "mobi")

    # producers of the shared inputs ---------------------------------
    def load_velocity(self, ocean):
        """Upload u(tau) and the metrics `adv_vel` needs."""
        g = ocean.grid
        self.upload("u1", np.asfortranarray(ocean.u[..., 0]))
        self.upload("u2", np.asfortranarray(ocean.u[..., 1]))
        self.upload("dxt2r", g.dxt2r)
        self.upload("dyt2r", g.dyt2r)

    def adv_vel(self):
        """adv_vet, adv_vnt, adv_vbt from u(tau) on the device (source/mom/adv_vel.F:63-131)."""
        check(self.lib.uvic_gpu_adv_vel(self.h), "adv_vel")

    def load_tidal(self, ocean, tidal):
        """Upload the inputs of the tidal-mixing scheme (updates/09/source/mom/tidal_kv.h)."""
        from .capi import VmixParams
        g = ocean.grid
        self.upload("zw", g.zw)
        self.upload("tlat", np.asfortranarray(tidal.tlat))
        for n in ("edrm2", "edrs2", "edrk1", "edro1"):
            self.upload(n, np.asfortranarray(getattr(tidal, n)))
        p = VmixParams(tidal.kappa_h, tidal.zetar, tidal.ogamma, tidal.gravrho0r)
        check(self.lib.uvic_gpu_set_vmix_params(self.h, ctypes.byref(p)), "set_vmix_params")

    def vmixc(self):
        """diff_cbt = tidal mixing + K33 on the device (updates/09/source/mom/vmixc.F:62-190); call after
        isopyc() with set_params(diff_cbt_has_k33=1)."""
        check(self.lib.uvic_gpu_vmixc(self.h), "vmixc")

    # baroclinic momentum step ------------------------------------------
    def load_momentum(self, ocean, mom):
        """Upload what `clinic` reads and the tracer step does not: u(tau), u(tau-1), wind stress, the U-grid metrics,
        the factors of setmom.F and the viscosity coefficients of hmixc.F (`mom` as synthetic.make_momentum)."""
        from .capi import ClinicParams
        g, topo = ocean.grid, ocean.topo
        self.upload("u1", np.asfortranarray(ocean.u[..., 0]))
        self.upload("u2", np.asfortranarray(ocean.u[..., 1]))
        self.upload("um1", np.asfortranarray(mom.u_taum1[..., 0]))
        self.upload("um2", np.asfortranarray(mom.u_taum1[..., 1]))
        self.upload("kmu", topo.kmu)
        for n in ("dxu2r", "dxmetr", "duw", "due", "dyu2r", "dyu4r", "csur", "dus", "dun", "csudyu2r"):
            self.upload(n, getattr(g, n))
        for n in ("smf", "hr", "cori", "visc_ceu", "amc_north", "amc_south", "advmet", "am3", "am4"):
            self.upload(n, getattr(mom, n))
        p = ClinicParams(2.0 * mom.dtuv, mom.grav, mom.rho0r, mom.kappa_m, mom.cdbot)
        check(self.lib.uvic_gpu_set_clinic_params(self.h, ctypes.byref(p)), "set_clinic_params")

    def state(self):
        """rho from T and S of t(tau) (source/mom/state.F as called at loadmw.F:154)."""
        check(self.lib.uvic_gpu_state(self.h), "state")

    def clinic(self, accumulate_sbc=False, osegs=False, osege=False, rts=1.0):
        """Internal-mode velocities at tau+1 and zu (updates/09/source/mom/clinic.F); returns u(tau+1) as
        (imt,km,jmt,2) and zu (imt,jmt,2)."""
        flags = (1 if accumulate_sbc else 0) | (2 if osegs else 0) | (4 if osege else 0)
        check(self.lib.uvic_gpu_clinic(self.h, flags, float(rts)), "clinic")
        return np.stack([self.download("up1"), self.download("up2")], axis=-1), self.download("zu")

    def clinic_only(self, accumulate_sbc=False, osegs=False, osege=False, rts=1.0):
        """clinic() without the downloads."""
        flags = (1 if accumulate_sbc else 0) | (2 if osegs else 0) | (4 if osege else 0)
        check(self.lib.uvic_gpu_clinic(self.h, flags, float(rts)), "clinic")

    def state_async(self):
        """Queue `state` on the main stream and return (sync() waits)."""
        check(self.lib.uvic_gpu_state_async(self.h), "state_async")

    def clinic_async(self, accumulate_sbc=False, osegs=False, osege=False, rts=1.0):
        flags = (1 if accumulate_sbc else 0) | (2 if osegs else 0) | (4 if osege else 0)
        check(self.lib.uvic_gpu_clinic_async(self.h, flags, float(rts)), "clinic_async")

    def set_host_sync(self, on):
        """False: uploads are queued on the main stream and return (the host buffer must stay as it is until the copy
        has run); True (default): they return when the copy is done."""
        check(self.lib.uvic_gpu_set_host_sync(self.h, 1 if on else 0), "set_host_sync")

    def rotate_u(self):
        """tau-1 <- tau <- tau+1 of the velocities, by pointer (what loadmw does with the memory-window slots)."""
        check(self.lib.uvic_gpu_rotate_u(self.h), "rotate_u")

    def add_ext_mode(self, level=0):
        """u(tau) (level 0) or u(tau-1) (level -1) += the external mode of UVIC_F_PSI (loadmw.F add_ext_mode)."""
        check(self.lib.uvic_gpu_add_ext_mode(self.h, int(level)), "add_ext_mode")

    def set_filter_u(self, ocean, flt):
        """Polar Fourier filter of u(tau+1) inside clinic (source/common/filuv.F); `flt` as synthetic.make_filter_u."""
        if flt is None:
            check(self.lib.uvic_gpu_set_filter_u(self.h, float(ocean.grid.pi), self.jmt + 1, 1, 1, 2, 1), "set_filter_u")
        else:
            self.upload("spsin", flt.spsin)
            self.upload("spcos", flt.spcos)
            self.upload("phi", ocean.grid.phi)
            check(self.lib.uvic_gpu_set_filter_u(self.h, float(ocean.grid.pi), flt.jfrst, flt.jfu0, flt.jfu1, flt.jfu2, flt.lsegf),
                  "set_filter_u")

    def set_filter(self, ocean, flt):
        """Polar Fourier filter of t(tau+1) after convection (source/common/filt.F); `flt` as
        synthetic.make_filter.  Call after load_ocean.  flt=None switches it off."""
        if flt is None:
            check(self.lib.uvic_gpu_set_filter(self.h, float(ocean.grid.pi), self.jmt + 1, 1, 1, 2, 1), "set_filter")
        else:
            check(self.lib.uvic_gpu_set_filter(self.h, float(ocean.grid.pi), flt.jfrst, flt.jft0, flt.jft1, flt.jft2, flt.lsegf),
                  "set_filter")

    def set_shard(self, n0=0, nt_local=None, js=2, je=None):
        nt_local = self.nt - n0 if nt_local is None else nt_local
        je = self.jmt - 1 if je is None else je
        check(self.lib.uvic_gpu_set_shard(self.h, n0, nt_local, js, je), "set_shard")

    def load_ocean(self, ocean, to, so, c, src=None):
        """Upload every static and per-step input from a synthetic Ocean."""
        g, topo, prm, cfg = ocean.grid, ocean.topo, ocean.params, ocean.cfg
        for n in _GRID_1D:
            self.upload(n, getattr(g, n))
        self.upload("to", to)
        self.upload("so", so)
        self.upload("c", c)
        self.upload("kmt", topo.kmt)
        self.upload("fisop", ocean.fisop)
        self.upload("addisop", ocean.addisop)
        self.upload("t_taum1", ocean.t_taum1)
        self.upload("t_tau", ocean.t_tau)
        self.upload("adv_vet", ocean.adv_vet)
        self.upload("adv_vnt", ocean.adv_vnt)
        self.upload("adv_vbt", ocean.adv_vbt)
        self.upload("diff_cbt_bg", ocean.diff_cbt_bg)
        self.upload("stf", ocean.stf)
        self.upload("btf", ocean.btf)
        self.upload("itrc", np.array(cfg.itrc(), dtype=np.int32))
        if src is not None:
            self.upload("src", src)
        self.set_params(c2dtts=2.0 * prm.dtts, aidif=prm.aidif, diff_cet=prm.diff_cet, diff_cnt=prm.diff_cnt,
                        slmxr=1.0 / prm.slmx, ahisop=prm.ahisop, athkdf=prm.athkdf)

    # -- the reference's entry points ------------------------------------------------
    def _check_window(self, who, joff, js, je, is_, ie, js_expected):
        if js > je:            # `if (js .gt. je) return`, tracer.F:219 / vmixc etc.
            return False
        if joff != 0 or is_ != 2 or ie != self.imt - 1 or js != js_expected[0] or je != js_expected[1]:
            # the reference runs one memory window covering all rows (jmw = jmt,
            # updates/09/source/common/size.h:155)
            raise UvicGpuError(f"{who}: only the single full memory window is supported "
                               f"(joff=0, js={js_expected[0]}, je={js_expected[1]}, is=2, ie=imt-1)")
        return True

    def prefetch_isopyc(self):
        """The T,S-derived fields of the next (leapfrog) step on a side stream, beside this step."""
        check(self.lib.uvic_gpu_prefetch_isopyc(self.h), "prefetch_isopyc")

    def isopyc(self, joff=0, js=1, je=None, is_=2, ie=None):
        je = self.jmt if je is None else je
        ie = self.imt - 1 if ie is None else ie
        if self._check_window("isopyc", joff, js, je, is_, ie, (1, self.jmt)):
            check(self.lib.uvic_gpu_isopyc(self.h), "isopyc")

    def tracer(self, joff=0, js=2, je=None, is_=2, ie=None):
        je = self.jmt - 1 if je is None else je
        ie = self.imt - 1 if ie is None else ie
        if self._check_window("tracer", joff, js, je, is_, ie, (2, self.jmt - 1)):
            check(self.lib.uvic_gpu_tracer(self.h), "tracer")

    def transport(self):
        check(self.lib.uvic_gpu_transport(self.h), "transport")

    def convect(self):
        check(self.lib.uvic_gpu_convect(self.h), "convect")

    def step_async(self):
        check(self.lib.uvic_gpu_step_async(self.h), "step_async")

    def prefetch_sources(self, c2dtts_next):
        check(self.lib.uvic_gpu_prefetch_sources(self.h, float(c2dtts_next)), "prefetch_sources")

    def set_exact(self, on):
        """Arithmetic of the transport step: False / 0 = production (T and S through the bit-exact kernels -- every
        convective adjustment is decided on their bits --, the other tracers through the column kernels);
        True / 1 = every tracer through the bit-exact kernels; "columns" / 2 = every tracer through the column kernels
        (T and S then agree with the reference to rounding only); "rows" / 3 = as production with T and S through the
        row kernels of kernels_fct.hpp instead of the exact column kernels (cross-check)."""
        mode = {"columns": 2, "rows": 3}.get(on) if isinstance(on, str) else (int(on) if on in (2, 3) and on is not True else (1 if on else 0))
        check(self.lib.uvic_gpu_set_exact(self.h, mode), "set_exact")

    def set_option(self, name, value):
        """cross-check and tuning switches of the library (include/uvic_gpu.h: uvic_gpu_set_option)"""
        check(self.lib.uvic_gpu_set_option(self.h, name.encode(), int(value)), f"set_option {name}")

    def last_error(self):
        return self.lib.uvic_gpu_last_error().decode()

    def set_mixing(self, on):
        check(self.lib.uvic_gpu_set_mixing(self.h, 1 if on else 0), "set_mixing")

    def rotate(self):
        check(self.lib.uvic_gpu_rotate(self.h), "rotate")

    def sync(self):
        check(self.lib.uvic_gpu_sync(self.h), "sync")

    def profile(self, nrep=5):
        names = (ctypes.c_char_p * 32)()
        ms = (ctypes.c_double * 32)()
        n = ctypes.c_int()
        check(self.lib.uvic_gpu_profile(self.h, nrep, 32, names, ms, ctypes.byref(n)), "profile")
        return {names[i].decode(): ms[i] for i in range(n.value)}


    def profile_live(self, on=True):
        """Bracket every kernel of the following steps with HIP events (see profile_read)."""
        check(self.lib.uvic_gpu_profile_live(self.h, 1 if on else 0), "profile_live")

    def profile_read(self):
        names = (ctypes.c_char_p * 32)()
        ms = (ctypes.c_double * 32)()
        n = ctypes.c_int()
        check(self.lib.uvic_gpu_profile_read(self.h, 32, names, ms, ctypes.byref(n)), "profile_read")
        return {names[i].decode(): ms[i] for i in range(n.value)}


class OceanLoop:
    """The memory-window loop of `mom` (/root/reference/source/mom/mom.F:289-408) with everything but `tropic` on the
    device: per time step the stream function goes up (one plane) and the vertically averaged forcing zu comes back (two
    planes); tracers and velocities stay and rotate there.

        loadmw   -> add_ext_mode (u(tau) += external mode of psi), state (rho from T,S at tau)
        adv_vel, isopyc (+K33), [setvbc: bottom drag inside clinic], tracer || clinic (clinic does not read what the
        tracer step writes: it runs on a stream of its own beside it, and zu is back on the host long before the
        tracer step ends)
        (host: tropic solves for the next psi from zu -- beside the tracer step)

    The tracer part is TimeLoop's step (look-ahead chains of the next step included: neither the MOBI sources nor the
    T,S-derived fields depend on the velocities; the total advective velocities a chain formed are redone from the new
    adv_vel).  Leapfrog steps only.  The caller has uploaded the ocean (load_ocean, load_velocity), the momentum inputs
    (load_momentum: internal-mode u(tau), u(tau-1)) and the two polar filters if wanted."""

    def __init__(self, model, dtts, dtuv, segment=0):
        self.m, self.dtts, self.dtuv = model, float(dtts), float(dtuv)
        self.tl = TimeLoop(model, dtts, nmix=0, segment=segment)
        self.itt = 0
        self.psi = np.zeros(model.shape("psi"), order="F")
        self.zu = np.zeros(model.shape("zu"), order="F")
        for a in (self.zu, self.psi):        # both cross PCIe every step: page-locked, copies queued without waiting
            check(model.lib.uvic_gpu_pin_host(model.h, a.ctypes.data_as(ctypes.c_void_p), a.nbytes), "pin_host")
        # the registration ends with the model (uvic_gpu_destroy): keep the arrays at least that long
        model._pinned = getattr(model, "_pinned", []) + [self.zu, self.psi]
        model.set_host_sync(False)

    def step(self, psi_tau, psi_taum1=None, accumulate_sbc=False, osegs=False, osege=False, rts=1.0):
        """One leapfrog step; psi_tau (imt,jmt): stream function at tau.  On the first step the reference adds the
        external mode to u(tau-1) as well (loadmw.F:88-90): give psi_taum1.  Returns zu (imt,jmt,2)."""
        m = self.m
        self.itt += 1
        self.psi[..., 0] = psi_tau
        if psi_taum1 is not None:
            self.psi[..., 1] = psi_taum1
        m.upload("psi", self.psi)
        m.add_ext_mode(0)
        if psi_taum1 is not None:
            m.add_ext_mode(-1)
        check(m.lib.uvic_gpu_adv_vel_async(m.h), "adv_vel_async")
        # state + clinic on their own stream, beside the tracer step; zu lands in page-locked memory
        flags = (1 if accumulate_sbc else 0) | (2 if osegs else 0) | (4 if osege else 0)
        check(m.lib.uvic_gpu_momentum_async(m.h, flags, float(rts), self.zu.ctypes.data_as(ctypes.c_void_p)), "momentum_async")
        self.tl.step()                       # isopyc, tracer, look-ahead chains, rotation of t: queued, not waited for
        m.rotate_u()
        check(m.lib.uvic_gpu_momentum_wait(m.h), "momentum_wait")   # the step's only wait: zu, not the tracer step
        return self.zu


class TimeLoop:
    """The ocean time-step schedule of `mom` for a device-resident model
    (/root/reference/source/mom/mom.F:108-148): leapfrog steps with c2dtts = 2*dtts and a
    forward "mixing" step with c2dtts = dtts every nmix-th step
    (/root/reference/updates/09/source/common/switch.F:217-223), on which t(tau-1) := t(tau).
    With MOBI, the source terms of the next leapfrog step are started one step ahead on a
    side stream (they depend only on t(tau-1) of that step = t(tau) of this one)."""

    def __init__(self, model, dtts, nmix=16, shard=None, prefetch=True, segment=0, clock=None, iso2=False):
        """segment = ntspos, the ocean steps per coupling segment (u09/common/UVic_ESCM.F:177-189): the surface forcing
        MOBI reads changes at a segment's first step, whose sources therefore cannot be computed one step ahead
        (0: the forcing never changes during the loop)."""
        self.m, self.dtts, self.nmix, self.shard, self.prefetch = model, float(dtts), int(nmix), shard, prefetch
        self.segment = int(segment)
        # clock = (relyr of the first step, increment per step, co2ccn): the reference advances relyr every ocean step and
        # MOBI takes the month of the dust field and the declination from it (u09/mom/tracer.F:311-338); None: it stands still
        self.clock = clock
        self.iso2 = bool(iso2)      # isopyc two steps ahead on the idle MOBI stream (measured: no gain)
        self.itt = 0

    def _mixing(self, itt):
        return self.nmix > 0 and itt % self.nmix == 0

    def step(self):
        m = self.m
        self.itt += 1
        mixing = self._mixing(self.itt)
        if self.shard is None or hasattr(self.shard, "after_step"):
            # one C call per step (uvic_gpu_step_lookahead), then the halo exchange of a latitude slab, then the rotation
            ahead = self.prefetch and not self._mixing(self.itt + 1)
            mobi_ahead = ahead and m.has_mobi and not (self.segment > 0 and self.itt % self.segment == 0)
            # bit 0: T,S-derived fields of the next step from t(tau); bit 1: those of the step after next from this step's
            # t(tau+1) (single rank only: a latitude slab gets the halo rows of t(tau+1) with the exchange that follows)
            iso_ahead = 0
            if self.prefetch and not m.params.diff_cbt_has_k33:
                iso_ahead = ((1 if ahead else 0) | (2 if self.shard is None and self.iso2 and not self._mixing(self.itt + 2) else 0)
                             | (4 if self.shard is None else 0))   # 4: nothing follows on the main stream before the next step
            c2dtts = self.dtts if mixing else 2.0 * self.dtts
            if self.clock is None:
                check(m.lib.uvic_gpu_step_lookahead(m.h, c2dtts, int(mixing), int(mobi_ahead), 2.0 * self.dtts, int(iso_ahead)),
                      "step_lookahead")
            else:
                relyr0, dyr, co2 = self.clock
                relyr = relyr0 + (self.itt - 1) * dyr
                if m.has_mobi:      # this step's clock (forcing fields unchanged), then the step with the next step's named
                    check(m.lib.uvic_gpu_set_mobi_step(m.h, relyr, co2, None, None, None, None), "set_mobi_step")
                check(m.lib.uvic_gpu_step_lookahead_at(m.h, c2dtts, int(mixing), int(mobi_ahead), 2.0 * self.dtts,
                                                       relyr0 + self.itt * dyr, co2, int(iso_ahead)), "step_lookahead_at")
            m.params.c2dtts = c2dtts          # the mirror of uvic_params follows
            if self.shard is not None:
                self.shard.after_step(m)
            m.rotate()
            return
        m.set_mixing(mixing)
        m.set_params(c2dtts=self.dtts if mixing else 2.0 * self.dtts)
        if self.shard is not None:
            self.shard.step(m)
        else:
            m.step_async()
        # the look-ahead chains of the next step wait only for the end of the previous step; they are queued after
        # this step so that its T,S passes (first on the isopyc stream) are not held up behind them
        if self.prefetch and not self._mixing(self.itt + 1):
            if m.has_mobi and not (self.segment > 0 and self.itt % self.segment == 0):   # next step opens a segment: new forcing
                m.prefetch_sources(2.0 * self.dtts)
            if not m.params.diff_cbt_has_k33:
                m.prefetch_isopyc()
        m.rotate()
        if mixing:
            m.set_mixing(False)
