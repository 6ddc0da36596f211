"""Deterministic synthetic ocean for tests and bench.

The reference's data directory (grid, topography, forcing, initial conditions)
is not part of the repository (run/mk.in:196), so every input of the tracer
step is generated from closed formulas.  Everything here is *input* of the
hot path: grid metrics as /root/reference/source/common/grids.F:416-563
derives them from the grid spacing, land/sea index arrays, initial tracer
profiles after /root/reference/updates/09/source/mom/setmom.F:1258-1340 and
:1530-1590, smooth velocities, and the advective velocities on T-cell faces as
/root/reference/source/mom/adv_vel.F:63-131 builds them from `u`.

All arrays are numpy, Fortran order, **1-based reference index i <-> python
index i-1**, with the package-wide layout `(imt, km, jmt[, ...])`; vertical
face fields carry `km+1` levels (reference `0:km`).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from types import SimpleNamespace

import numpy as np

from .configs import OPTION_SETS, OptionSet

RADIUS = 6370.0e5          # cm, updates/09/source/common/UVic_ESCM.F:1647
RN15STD = 0.0036765        # updates/09/source/mom/mobi.h:199-212
RC13STD = 0.0112372
RC14STD = 1.176e-12


def F(shape, dtype=np.float64):
    return np.zeros(shape, dtype=dtype, order="F")


def make_grid(imt: int, jmt: int, km: int):
    """Uniform lat-lon grid, `dzt(k)` growing linearly with depth."""
    g = SimpleNamespace(imt=imt, jmt=jmt, km=km)
    pi = 4.0 * np.arctan(1.0)
    radian = 360.0 / (2.0 * pi)
    degtcm = RADIUS / radian
    g.pi, g.radian = pi, radian
    dx = 360.0 / (imt - 2)
    dy = 180.0 / (jmt - 2)
    i = np.arange(1, imt + 1, dtype=np.float64)
    j = np.arange(1, jmt + 1, dtype=np.float64)
    g.dxtdeg = np.full(imt, dx)
    g.dxudeg = np.full(imt, dx)
    g.dytdeg = np.full(jmt, dy)
    g.dyudeg = np.full(jmt, dy)
    g.xt = (i - 1.5) * dx
    g.xu = (i - 1.0) * dx
    g.yt = -90.0 + (j - 1.5) * dy
    g.yu = -90.0 + (j - 1.0) * dy
    if km == 19:
        dzt = 50.0e2 + 30.0e2 * np.arange(km)
    else:
        dzt = 30.0e2 + 10.0e2 * np.arange(km)
    g.dzt = dzt.astype(np.float64)
    g.zw = np.cumsum(g.dzt)
    g.zt = g.zw - 0.5 * g.dzt
    dzw = np.zeros(km + 1)
    dzw[0] = g.zt[0]
    dzw[1:km] = g.zt[1:] - g.zt[:-1]
    dzw[km] = g.zw[km - 1] - g.zt[km - 1]
    g.dzw = dzw
    # metrics, grids.F:416-563
    g.dyt = g.dytdeg * degtcm
    g.dyu = g.dyudeg * degtcm
    g.dxt = g.dxtdeg * degtcm
    g.dxu = g.dxudeg * degtcm
    g.c2dzt = 2.0 * g.dzt
    g.dzt2r = 1.0 / g.c2dzt
    g.dzwr = 1.0 / g.dzw
    g.dzw2r = 0.5 / g.dzw
    g.dztur = 1.0 / (g.dzw[0:km] * g.dzt)
    g.dztlr = 1.0 / (g.dzw[1:km + 1] * g.dzt)
    g.dztr = 1.0 / g.dzt
    g.dytr = 1.0 / g.dyt
    g.dyt2r = 0.5 / g.dyt
    g.dyt4r = 0.25 / g.dyt
    g.dyur = 1.0 / g.dyu
    g.dyu2r = 0.5 / g.dyu
    g.dyu4r = 0.25 / g.dyu
    g.phi = g.yu / radian
    g.phit = g.yt / radian
    g.cst = np.cos(g.phit)
    g.csu = np.cos(g.phi)
    g.sine = np.sin(g.phi)
    g.cst[g.cst == 0.0] = 1.0e-20
    g.csu[g.csu == 0.0] = 1.0e-20
    # the pole rows of the U grid: cos(+-90 deg) is ~6e-17, keep as computed
    g.cstr = 1.0 / g.cst
    g.csur = 1.0 / g.csu
    g.tng = g.sine / g.csu
    g.cstdytr = 1.0 / (g.cst * g.dyt)
    g.cstdyt2r = g.cstdytr * 0.5
    g.csudyur = 1.0 / (g.csu * g.dyu)
    g.csudyu2r = 0.5 / (g.csu * g.dyu)
    g.cst_dytr = g.cst / g.dyt
    g.csu_dyur = g.csu / g.dyu
    g.dxtr = 1.0 / g.dxt
    g.dxt2r = 0.5 / g.dxt
    g.dxt4r = 0.25 / g.dxt
    g.dxur = 1.0 / g.dxu
    g.dxu2r = 0.5 / g.dxu
    g.dxu4r = 0.25 / g.dxu
    g.dtxcel = np.ones(km)
    g.dtxsqr = np.sqrt(g.dtxcel)
    g.dztxcl = g.dzt / g.dtxcel
    dzwxcl = np.zeros(km)
    dzwxcl[:km - 1] = 1.0 / (g.dztxcl[:-1] + g.dztxcl[1:])
    g.dzwxcl = dzwxcl
    # U-cell metrics of the momentum equations, grids.F:527-550 (cyclic)
    g.dxmetr = np.zeros(imt)
    g.dxmetr[1:imt - 1] = 1.0 / (g.dxt[1:imt - 1] + g.dxt[2:imt])
    g.duw = (g.xu - g.xt) * degtcm
    g.due = np.zeros(imt)
    g.due[:imt - 1] = (g.xt[1:] - g.xu[:imt - 1]) * degtcm
    g.due[imt - 1] = g.due[1]
    g.dus = (g.yu - g.yt) * degtcm
    g.dun = np.zeros(jmt)
    g.dun[:jmt - 1] = (g.yt[1:] - g.yu[:jmt - 1]) * degtcm
    g.dun[jmt - 1] = g.dun[jmt - 2]
    g.tlat = F((imt, jmt))
    g.tlat[:, :] = g.yt[None, :]
    g.tlon = F((imt, jmt))
    g.tlon[:, :] = g.xt[:, None]
    return g


def make_topography(g):
    """kmt/kmu (int32, bit-exact data), tmask/umask, sg_bathy.

    Land poleward of +-72 deg, two meridional land strips, depth varying in
    [km-5, km] with a few shallow shelf columns (kmt = 1, 2, 3)."""
    imt, jmt, km = g.imt, g.jmt, g.km
    kmt = np.zeros((imt, jmt), dtype=np.int32, order="F")
    for j in range(2, jmt):          # 1-based rows 2..jmt-1
        lat = g.yt[j - 1]
        for i in range(2, imt):
            k = km - ((3 * i + 5 * j) % 6)
            if abs(lat) > 72.0:
                k = 0
            fi = (i - 2) / float(imt - 2)
            if 0.18 <= fi < 0.23 and -50.0 < lat < 65.0:
                k = 0
            if 0.58 <= fi < 0.66 and -35.0 < lat < 72.0:
                k = 0
            if k > 0 and (0.23 <= fi < 0.25) and -50.0 < lat < 65.0:
                k = 1 + ((i + j) % 3)        # shelf: 1, 2 or 3 levels
            if k > 0 and (0.56 <= fi < 0.58) and -35.0 < lat < 72.0:
                k = max(2, km // 3)
            kmt[i - 1, j - 1] = min(k, km)
    kmt[0, :] = kmt[imt - 2, :]
    kmt[imt - 1, :] = kmt[1, :]
    kmu = np.zeros_like(kmt)
    kmu[:imt - 1, :jmt - 1] = np.minimum(np.minimum(kmt[:imt - 1, :jmt - 1], kmt[1:, :jmt - 1]),
                                         np.minimum(kmt[:imt - 1, 1:], kmt[1:, 1:]))
    kmu[0, :] = kmu[imt - 2, :]
    kmu[imt - 1, :] = kmu[1, :]
    lev = np.arange(1, km + 1)[None, :, None]
    tmask = np.asfortranarray((kmt[:, None, :] >= lev).astype(np.float64))
    umask = np.asfortranarray((kmu[:, None, :] >= lev).astype(np.float64))
    sg = F((imt, jmt, km))
    ii, jj = np.nonzero(kmt > 0)
    sg[ii, jj, kmt[ii, jj] - 1] = 1.0
    return SimpleNamespace(kmt=kmt, kmu=kmu, tmask=tmask, umask=umask, sg_bathy=sg)


def _profile(name: str, g, k0: int) -> float:
    """Initial value of tracer `name` at level k0 (0-based), setmom.F:1258-1340."""
    zt = g.zt
    k = k0 + 1
    e100 = np.exp((zt[0] - zt[k0]) / 100.0e2)
    table = {
        "dic": 2.315, "alk": 2.429, "sil": 0.084, "o2": 0.1692,
        "po4": 0.543 if k == 1 else 2.165,
        "dop": 0.156 if k <= 2 else (0.039 if k <= 12 else 0.0078),
        "phyt": 0.14 * e100, "phyt_phos": 0.14 * e100 * (1.0 / 16.0),
        "zoop": 0.014 * e100, "detr": 1.0e-4, "detr_phos": 1.0e-4 * (1.0 / 16.0),
        "detrfe": 1.0e-4 * 14.0e-6 * 6.625,
        "no3": 5.30 if k == 1 else 30.84, "dfe": 0.6e-3,
        "don": 3.5 if k <= 2 else (1.5 if k <= 12 else 0.5),
        "diaz": 0.014 * e100, "c14": -150.0, "caco3": 5.0e-5,
        "diat": 0.14 * e100, "opl": 5.0e-5,
    }
    if name in table:
        return table[name]
    return 1.0 if k == 1 else 0.0


def make_tracers(cfg: OptionSet, g, topo, phase: float = 0.0):
    """One time level of all tracers, shape (imt, km, jmt, nt)."""
    imt, jmt, km = g.imt, g.jmt, g.km
    nt = cfg.nt
    t = F((imt, km, jmt, nt))
    lam = 2.0 * np.pi * (np.arange(1, imt + 1) - 2.0) / (imt - 2)
    phi = g.phit
    z = g.zt
    pert = 1.0 + 0.1 * np.sin(lam + phase)[:, None, None] * np.cos(3.0 * phi)[None, None, :] \
        * (1.0 + 0.3 * np.cos(2.0 * np.pi * np.arange(km) / km + phase))[None, :, None]
    for n, name in enumerate(cfg.tracers):
        if name == "temp":
            base = 2.0 + 23.0 * (np.cos(phi) ** 2)[None, None, :] * np.exp(-z / 800.0e2)[None, :, None]
            f = base * (1.0 + 0.02 * (pert - 1.0))
            # a cold surface patch makes some columns statically unstable (convct2)
            f = f - 6.0 * np.exp(-((phi - 1.0) / 0.12) ** 2)[None, None, :] \
                * np.exp(-z / 150.0e2)[None, :, None] * (0.5 + 0.5 * np.sin(lam))[:, None, None]
        elif name == "salt":
            s_psu = 34.7 + 0.6 * (np.cos(phi) ** 2)[None, None, :] * np.exp(-z / 1000.0e2)[None, :, None]
            f = (s_psu - 35.0) * 1.0e-3 * (1.0 + 0.05 * (pert - 1.0))
        elif name.startswith("trc"):
            prof = np.array([1.0 + np.exp(-z[k] / 500.0e2) * (1 + (n % 5)) for k in range(km)])
            f = prof[None, :, None] * pert
        else:
            prof = np.array([_profile(name, g, k) for k in range(km)])
            f = prof[None, :, None] * pert
        t[:, :, :, n] = f
    ix = {name: n for n, name in enumerate(cfg.tracers)}
    # isotopes, setmom.F:1530-1590
    if "din15" in ix:
        t[..., ix["din15"]] = 1.005 * RN15STD * t[..., ix["no3"]] / (1 + 1.005 * RN15STD)
        t[..., ix["don15"]] = RN15STD * t[..., ix["don"]] / (1 + RN15STD)
        for a, b in (("phytn15", "phyt"), ("zoopn15", "zoop"), ("detrn15", "detr"), ("diazn15", "diaz")):
            t[..., ix[a]] = RN15STD * t[..., ix[b]] / (1 + RN15STD)
    if "dic13" in ix:
        t[..., ix["dic13"]] = t[..., ix["dic"]] * 1.0004 * RC13STD / (1.0 + 1.0004 * RC13STD)
        for a, b in (("phytc13", "phyt"), ("zoopc13", "zoop"), ("detrc13", "detr"),
                     ("doc13", "don"), ("diazc13", "diaz")):
            if a in ix:
                t[..., ix[a]] = t[..., ix[b]] * 6.625e-3 * RC13STD / (1.0 + RC13STD)
    if "c14" in ix:
        t[..., ix["c14"]] = (t[..., ix["c14"]] * 0.001 + 1) * t[..., ix["dic"]] * RC14STD
    t *= topo.tmask[:, :, :, None]
    t[0] = t[imt - 2]
    t[imt - 1] = t[1]
    return t


def make_velocity(g, topo):
    """Smooth horizontal velocity on the U grid, cm/s, shape (imt, km, jmt, 2)."""
    imt, jmt, km = g.imt, g.jmt, g.km
    u = F((imt, km, jmt, 2))
    lam = 2.0 * np.pi * (np.arange(1, imt + 1) - 1.5) / (imt - 2)
    phi = g.phi
    ez = np.exp(-g.zt / 1000.0e2)
    u[..., 0] = 3.0 * np.cos(phi)[None, None, :] * ez[None, :, None] * (1.0 + 0.5 * np.sin(2.0 * lam))[:, None, None]
    u[..., 1] = 1.5 * np.sin(2.0 * lam + 0.3)[:, None, None] * np.sin(2.0 * phi)[None, None, :] \
        * np.exp(-g.zt / 800.0e2)[None, :, None]
    u *= topo.umask[..., None]
    u[0] = u[imt - 2]
    u[imt - 1] = u[1]
    return u


def make_momentum(g, topo, u_tau, anisotropic=True):
    """Inputs of the baroclinic momentum step (`clinic`) that the tracer step does not
    have: u(tau-1), wind stress, the static factors of setmom.F:770-803, 1104-1114 and the viscosity
    coefficients hmixc.F leaves on its first call (O_anisotropic_viscosity: three 3-D fields; here a smooth
    synthetic pattern in place of hmixc.F:49-103's western-boundary rule, amc_* from it as hmixc.F:120-131)."""
    imt, jmt, km = g.imt, g.jmt, g.km
    m = SimpleNamespace()
    lam = 2.0 * np.pi * (np.arange(1, imt + 1) - 1.0) / (imt - 2)
    # u(tau-1): the tau field, slightly rotated and damped
    um = F((imt, km, jmt, 2))
    um[..., 0] = 0.97 * u_tau[..., 0] - 0.05 * u_tau[..., 1]
    um[..., 1] = 0.97 * u_tau[..., 1] + 0.05 * u_tau[..., 0]
    um *= topo.umask[..., None]
    um[0] = um[imt - 2]
    um[imt - 1] = um[1]
    m.u_taum1 = um
    m.smf = F((imt, jmt, 2))
    m.smf[..., 0] = 0.8 * np.cos(3.0 * g.phi)[None, :] * (1.0 + 0.2 * np.sin(lam))[:, None] * topo.umask[:, 0, :]
    m.smf[..., 1] = 0.1 * np.sin(2.0 * lam + 1.0)[:, None] * np.cos(g.phi)[None, :] * topo.umask[:, 0, :]
    m.smf[0] = m.smf[imt - 2]
    m.smf[imt - 1] = m.smf[1]
    omega = g.pi / 43082.0
    radius = RADIUS
    ulat = F((imt, jmt)); ulat[:, :] = g.yu[None, :]
    m.cori = F((imt, jmt, 2))
    m.cori[..., 0] = 2.0 * omega * np.sin(ulat / g.radian)
    m.cori[..., 1] = -m.cori[..., 0]
    m.am = 1.5e9
    m.am3 = m.am * (1.0 - g.tng * g.tng) / (radius ** 2)
    m.am4 = F((jmt, 2))
    m.am4[:, 0] = -m.am * 2.0 * g.sine / (radius * g.csu * g.csu)
    m.am4[:, 1] = -m.am4[:, 0]
    m.advmet = F((jmt, 2))
    m.advmet[:, 0] = g.tng / radius
    m.advmet[:, 1] = -m.advmet[:, 0]
    m.hr = F((imt, jmt))
    wet = topo.kmu > 0
    m.hr[wet] = 1.0 / g.zw[topo.kmu[wet] - 1]
    trop = (np.abs(g.yu) <= 20.0)[None, None, :] & (g.zw <= 55000.0)[None, :, None]
    visc_cnu = F((imt, km, jmt)); visc_cnu[:] = m.am
    visc_cnu[:] = np.where(trop, m.am * (1.0 + 2.0 * np.cos(lam)[:, None, None] ** 2), m.am)
    m.visc_ceu = F((imt, km, jmt))
    m.visc_ceu[:] = np.where(trop, m.am * (0.4 + 0.3 * np.sin(lam)[:, None, None] ** 2), m.am)
    jp1 = np.minimum(np.arange(jmt) + 1, jmt - 1)
    m.amc_north = F((imt, km, jmt)); m.amc_south = F((imt, km, jmt))
    m.amc_north[:] = visc_cnu * (g.cst[jp1] * g.dytr[jp1] * g.csur * g.dyur)[None, None, :]
    m.amc_south[:] = visc_cnu * (g.cst * g.dytr * g.csur * g.dyur)[None, None, :]
    if not anisotropic:
        # one coefficient per row (hmixc.F:104-136 without O_anisotropic_viscosity): visc_ceu = visc_cnu = am; the same
        # numbers spread over (imt,km,jmt) are what the device fields hold
        m.amc_north_row = m.am * (g.cst[jp1] * g.dytr[jp1] * g.csur * g.dyur)
        m.amc_south_row = m.am * (g.cst * g.dytr * g.csur * g.dyur)
        m.visc_ceu[:] = m.am
        m.amc_north[:] = m.amc_north_row[None, None, :]
        m.amc_south[:] = m.amc_south_row[None, None, :]
    m.anisotropic = anisotropic
    m.kappa_m, m.cdbot, m.dtuv = 10.0, 1.3e-3, 1125.0
    m.grav, m.rho0r = 980.6, 1.0 / 1.035
    return m


def advective_velocities(g, u):
    """adv_vet, adv_vnt (imt,km,jmt) and adv_vbt (imt,km+1,jmt) from `u`
    exactly as /root/reference/source/mom/adv_vel.F:63-131 (rigid lid)."""
    imt, jmt, km = g.imt, g.jmt, g.km
    vnt = F((imt, km, jmt))
    vet = F((imt, km, jmt))
    vbt = F((imt, km + 1, jmt))
    I = slice(1, imt - 1)            # i = 2..imt-1
    Im = slice(0, imt - 2)           # i-1
    # rows 1..jmt (python 0..jmt-1)
    vnt[I] = (u[I, :, :, 1] * g.dxu[I, None, None] + u[Im, :, :, 1] * g.dxu[Im, None, None]) \
        * g.csu[None, None, :] * g.dxt2r[I, None, None]
    vnt[0] = vnt[imt - 2]
    vnt[imt - 1] = vnt[1]
    # rows 2..jmt, i = 1..imt
    vet[:, :, 1:] = (u[:, :, 1:, 0] * g.dyu[None, None, 1:] + u[:, :, :-1, 0] * g.dyu[None, None, :-1]) \
        * g.dyt2r[None, None, 1:]
    div = F((imt, km, jmt))
    div[I, :, 1:] = ((vet[I, :, 1:] - vet[Im, :, 1:]) * g.dxtr[I, None, None]
                     + (vnt[I, :, 1:] - vnt[I, :, :-1]) * g.dytr[None, None, 1:]) \
        * g.cstr[None, None, 1:] * g.dzt[None, :, None]
    for k in range(1, km + 1):
        vbt[:, k, :] = div[:, k - 1, :] + vbt[:, k - 1, :]
    vbt[0] = vbt[imt - 2]
    vbt[imt - 1] = vbt[1]
    return vet, vnt, vbt


def make_tidal(g, topo, kappa_h=0.35):
    """Synthetic inputs of the tidal-mixing part of `vmixc` (updates/09/source/mom/tidal_kv.h,
    vmixc.F:84-122): smooth positive energy dissipation rates of the four constituents on the
    sub-grid bathymetry, the constants of setmom.F:80-82 and the T-cell latitude."""
    imt, jmt, km = g.imt, g.jmt, g.km
    lam = 2.0 * np.pi * (np.arange(1, imt + 1) - 1.5) / (imt - 2)
    shape = (1.0 + 0.6 * np.sin(3.0 * lam))[:, None, None] * (1.0 + 0.5 * np.cos(2.0 * g.phi))[None, None, :]
    prof = (0.2 + (g.zt / g.zt[-1]) ** 2)[None, :, None]       # more dissipation near the bottom
    base = 2.0e-2 * shape * prof * topo.tmask
    rho0r, grav = 1.0 / 1.035, 980.6
    zetar = 1.0 / 500.0e2
    return SimpleNamespace(tlat=g.tlat, edrm2=F_(base), edrs2=F_(0.45 * base), edrk1=F_(0.3 * base), edro1=F_(0.2 * base),
                           kappa_h=float(kappa_h), zetar=zetar, ogamma=0.2 * rho0r * zetar, gravrho0r=grav * rho0r)


def indp(value, array):
    """1-based index of the element of the increasing `array` nearest to `value`
    (/root/reference/source/common/util.F `indp`)."""
    a = np.asarray(array)
    if value < a[0]:
        return 1
    if value > a[-1]:
        return len(a)
    for i in range(1, len(a)):
        if value <= a[i]:
            return i if a[i] - value > value - a[i - 1] else i + 1
    return len(a)


def make_filter(g, km, lsegf=20, jmtfil=50):
    """Rows of the polar Fourier filter of the tracers from the latitudes of
    /root/reference/updates/09/source/common/setcom.F:36-40, 75-85 (rjfrst=-87.3, rjft0=-67.5,
    rjft1=-69.3, rjft2=69.3); lsegf, jmtfil as source/common/index.h:34."""
    yt = g.yt
    f = SimpleNamespace(jfrst=indp(-87.3, yt), jft0=indp(-67.5, yt), jft1=indp(-69.3, yt), jft2=indp(69.3, yt),
                        lsegf=lsegf, jmtfil=jmtfil, km=km)
    f.jskpt = f.jft2 - f.jft1
    f.njtbft = (f.jft1 - f.jfrst + 1) + (g.jmt - 1 - f.jft2 + 1)
    if f.njtbft > jmtfil:
        f.jmtfil = f.njtbft
    return f


def make_filter_u(g, km, lsegf=20, jmtfil=50):
    """Rows of the polar Fourier filter of the velocities (`filuv`): latitudes of
    /root/reference/updates/09/source/common/setcom.F:38-44, 76-86 (rjfrst=-87.3 on the T grid; rjfu0=-68.4,
    rjfu1=-70.2, rjfu2=70.2 on the U grid) and the rotation factors of source/common/setcom.F:55-70."""
    f = SimpleNamespace(jfrst=indp(-87.3, g.yt), jfu0=indp(-68.4, g.yu), jfu1=indp(-70.2, g.yu), jfu2=indp(70.2, g.yu),
                        lsegf=lsegf, jmtfil=jmtfil, km=km)
    if g.jmt < 30:     # the small test grids have land poleward of 70 degrees: move the bounds to rows with ocean
        f.jfu1, f.jfu2 = max(f.jfu1, 4), min(f.jfu2, g.jmt - 4)
        f.jfu0 = f.jfu1 + 1
    f.jskpu = f.jfu2 - f.jfu1
    f.njtbfu = (f.jfu1 - f.jfrst + 1) + (g.jmt - 1 - f.jfu2 + 1)
    if f.njtbfu > jmtfil:
        f.jmtfil = f.njtbfu
    imt = g.imt
    fxa = g.dxt[0] / RADIUS
    fxb = fxa * (np.arange(1, imt + 1, dtype=np.float64) - 2.0)
    f.spsin, f.spcos = np.sin(fxb), np.cos(fxb)
    f.spsin[np.abs(f.spsin) < 1.0e-10] = 0.0
    f.spcos[np.abs(f.spcos) < 1.0e-10] = 0.0
    for a in (f.spsin, f.spcos):
        a[0] = 0.0
        a[imt - 1] = 0.0
    return f


def F_(a):
    return np.asfortranarray(a, dtype=np.float64)


@dataclass
class Ocean:
    cfg: OptionSet
    grid: SimpleNamespace
    topo: SimpleNamespace
    t_taum1: np.ndarray
    t_tau: np.ndarray
    u: np.ndarray
    adv_vet: np.ndarray
    adv_vnt: np.ndarray
    adv_vbt: np.ndarray
    diff_cbt_bg: np.ndarray
    stf: np.ndarray
    btf: np.ndarray
    fisop: np.ndarray
    addisop: np.ndarray
    forcing: SimpleNamespace
    params: SimpleNamespace = field(default_factory=SimpleNamespace)


def make_ocean(cfg="c30", imt=102, jmt=102, km=19) -> Ocean:
    if isinstance(cfg, str):
        cfg = OPTION_SETS[cfg]
    g = make_grid(imt, jmt, km)
    topo = make_topography(g)
    t0 = make_tracers(cfg, g, topo, phase=0.0)
    t1 = make_tracers(cfg, g, topo, phase=0.35)
    u = make_velocity(g, topo)
    vet, vnt, vbt = advective_velocities(g, u)
    nt = cfg.nt
    # vertical diffusivity before K33 is added: constant background + bottom-enhanced part
    depth = np.where(topo.kmt > 0, g.zw[np.maximum(topo.kmt, 1) - 1], 0.0)
    hab = np.maximum(depth[:, None, :] - g.zw[None, :, None], 0.0)
    kappa_h = 0.35
    dcb = F((imt, km, jmt))
    dcb[:] = kappa_h + 3.0 * np.exp(-hab / 500.0e2) * (topo.kmt[:, None, :] > 0)
    stf = F((imt, jmt, nt))
    btf = F((imt, jmt, nt))
    surf = t0[:, 0, :, :]
    stf[:] = 2.0e-6 * np.abs(surf) * np.cos(g.phit)[None, :, None] * (topo.kmt[:, :, None] > 0)
    btf[:] = -1.0e-7 * np.abs(surf) * (topo.kmt[:, :, None] > 0)
    fisop = F((imt, jmt, km))
    fisop[:] = 1.0 + 0.25 * np.cos(2.0 * g.phit)[None, :, None]
    addisop = F((imt, km, jmt))
    lam = 2.0 * np.pi * (np.arange(1, imt + 1) - 2.0) / (imt - 2)
    addisop[:] = 2.0e6 * (np.abs(g.yt) < 10.0)[None, None, :] * (0.5 + 0.5 * np.cos(lam))[:, None, None]
    frc = SimpleNamespace()
    frc.dnswr = F((imt, jmt))
    frc.dnswr[:] = 1.5e5 * np.cos(g.phit)[None, :]
    icy = (np.abs(g.yt) > 60.0)[None, :]
    frc.aice = F((imt, jmt)); frc.aice[:] = 0.5 * icy
    frc.hice = F((imt, jmt)); frc.hice[:] = 100.0 * icy
    frc.hsno = F((imt, jmt)); frc.hsno[:] = 10.0 * icy
    frc.co2ccn = 280.0
    frc.relyr = 0.3
    frc.fe_atmdep = F((imt, jmt, 12))
    frc.fe_atmdep[:] = 2.0e-13 * np.cos(g.phit)[None, :, None] * (1.0 + 0.1 * np.arange(12))[None, None, :]
    frc.fe_hydr = F((imt, jmt, km))
    kk = np.arange(1, km + 1)[None, None, :]
    frc.fe_hydr[:] = 1.0e-13 * ((kk == np.maximum(topo.kmt[:, :, None] - 1, 1)) & (topo.kmt[:, :, None] > 3))
    prm = SimpleNamespace(dtts=108000.0, aidif=0.5, kappa_h=kappa_h, diff_cet=1.0e5, diff_cnt=1.0e5,
                          slmx=0.01, ahisop=1.2e7, athkdf=8.0e6, nmix=16)
    return Ocean(cfg, g, topo, t0, t1, u, vet, vnt, vbt, dcb, stf, btf, fisop, addisop, frc, prm)


def load_eos(km: int):
    """Equation-of-state reference profiles and polynomial coefficients
    `to(km), so(km), c(km,9)` (source/mom/state.h:38) for the synthetic vertical
    grid with `km` levels.  They are outputs of the reference's `eqstate`
    (source/mom/denscoef.F, initialisation -- out of scope),
    stored by tests/golden/make_golden.py in data/eos.json."""
    import json
    from pathlib import Path
    tab = json.loads((Path(__file__).resolve().parent / "data" / "eos.json").read_text())
    if str(km) not in tab:
        raise KeyError(f"no equation-of-state table for km={km}; available: {sorted(tab)}")
    e = tab[str(km)]
    to = np.array(e["to"], dtype=np.float64)
    so = np.array(e["so"], dtype=np.float64)
    c = np.asfortranarray(np.array(e["c"], dtype=np.float64).T)
    return to, so, c


def pad_tracers(ocean: Ocean, nt_model: int) -> Ocean:
    """Append inert (all-zero, source-free) tracers so that the tracer dimension
    is a multiple of the number of ranks (parallel.py)."""
    import copy
    from dataclasses import replace
    nt = ocean.cfg.nt
    extra = nt_model - nt
    if extra <= 0:
        return ocean
    cfg = OptionSet(ocean.cfg.name, ocean.cfg.tracers + tuple(f"pad{n}" for n in range(extra)),
                    ocean.cfg.sources, ocean.cfg.mobi, ocean.cfg.options)

    def pad(a, axis):
        shp = list(a.shape)
        shp[axis] = extra
        return np.asfortranarray(np.concatenate([a, np.zeros(shp)], axis=axis))
    return replace(copy.copy(ocean), cfg=cfg, t_taum1=pad(ocean.t_taum1, 3), t_tau=pad(ocean.t_tau, 3),
                   stf=pad(ocean.stf, 2), btf=pad(ocean.btf, 2))
