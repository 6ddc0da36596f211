"""Baroclinic momentum step: `state` + `clinic` with the U-cell part of `adv_vel`, the
bottom drag of `setvbc`, `isbcu`/`asbcu` and the polar filter `filuv`.

CPU: C restatement == compiled reference (configuration "m2" of oracle/build_ref.py), bit for bit; the same against
     the committed fixture tests/golden/clinic_m2_14x14x6.npz (outputs of the compiled reference); host-emulated
     kernels == C restatement, bit for bit.
GPU: library == C restatement, bit for bit (integer-exact order of operations, no contraction), at 14x14x6 and at
     BASELINE's 102x102x19; size-independent properties at 102x102x19 (no vertical mean left, land untouched,
     cyclic columns)."""
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
for p in (ROOT, ROOT / "oracle", ROOT / "tests" / "hostemu"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

GRIDS = [(14, 14, 6), (102, 102, 19)]


def _setup(imt, jmt, km, cfg="m2"):
    from uvic29_amd import synthetic
    oc = synthetic.make_ocean(cfg, imt, jmt, km)
    mom = synthetic.make_momentum(oc.grid, oc.topo, oc.u, anisotropic=(cfg == "m2"))
    return oc, mom, synthetic.load_eos(km), synthetic.make_filter_u(oc.grid, km)


def _oracle_step(oc, mom, eos, flt=None):
    import oracle_c
    rho = oracle_c.state(oc.grid, eos, oc.t_tau[..., 0], oc.t_tau[..., 1])
    M = oracle_c.Momentum(oc, mom, rho)
    up, zu = M.step()
    if flt is not None:
        up = oracle_c.filuv(up, oc.grid, oc.topo, mom, flt)
    return rho, M, up, zu


@pytest.mark.parametrize("cfg,imt,jmt,km", [("m2",) + g for g in GRIDS] + [("m2i", 14, 14, 6)])
@pytest.mark.parametrize("filtered", [False, True])
def test_oracle_equals_reference(cfg, imt, jmt, km, filtered):
    """(m2i: the reference built without O_anisotropic_viscosity and O_ice_evp -- one viscosity per row.)"""
    import oracle_c
    import refdriver
    import refmodel
    if not refmodel.available(cfg, imt, jmt, km):
        pytest.skip("oracle/_ref build %s %dx%dx%d not present" % (cfg, imt, jmt, km))
    oc, mom, eos, flt = _setup(imt, jmt, km, cfg)
    R = refdriver.RefOcean(oc)
    R.set_momentum(mom)
    if filtered:
        R.set_filter_u(flt)
    rho = oracle_c.state(oc.grid, eos, oc.t_tau[..., 0], oc.t_tau[..., 1])
    assert np.array_equal(rho, R.state())
    M = oracle_c.Momentum(oc, mom, rho)
    for got, want in zip(M.adv_vel_u(), R.adv_vel_u()):
        assert np.array_equal(got, want)
    smf, bmf = R.setvbc()
    assert np.array_equal(smf, mom.smf) and np.array_equal(M.bmf(), bmf)
    up_ref, zu_ref, gp_ref = R.clinic()
    up, zu = M.clinic()
    if filtered:
        up = oracle_c.filuv(up, oc.grid, oc.topo, mom, flt)
    assert np.array_equal(M.a["grad_p"], gp_ref)
    assert np.array_equal(zu, zu_ref)
    assert np.array_equal(up, up_ref)
    assert np.abs(up).max() > 0.1 and np.abs(zu).max() > 1e-5
    if not filtered:    # loadmw's add_ext_mode, both time levels
        psi = _psi(oc.grid, 1)
        assert np.array_equal(M.add_ext_mode(psi, oc.u), R.add_ext_mode(psi, "tau"))
        assert np.array_equal(M.add_ext_mode(psi, mom.u_taum1), R.add_ext_mode(psi, "tau-1"))


def test_oracle_sbc_accumulation_equals_reference():
    import oracle_c
    import refdriver
    import refmodel
    if not refmodel.available("m2", 14, 14, 6):
        pytest.skip("oracle/_ref build m2 14x14x6 not present")
    oc, mom, eos, _ = _setup(14, 14, 6)
    R = refdriver.RefOcean(oc)
    R.set_momentum(mom)
    M = oracle_c.Momentum(oc, mom, R.state())
    S, v = R.ref.set, R.v
    S("igu", 11); S("igv", 12); S("isu", 13); S("isv", 14); S("ntspos", 3)
    rng = np.random.default_rng(0)
    for p in (10, 11, 12, 13):
        v["sbc"][:, :, p] = rng.standard_normal((14, 14))
    for osegs, osege in ((1, 0), (0, 0), (0, 1), (1, 1)):
        S("osegs", osegs); S("osege", osege)
        planes = [np.array(v["sbc"][:, :, p], order="F") for p in (10, 11, 12, 13)]
        R.clinic()
        M.sbcu("i", planes[0], planes[1], osegs, osege, 1.0 / 3)
        M.sbcu("a", planes[2], planes[3], osegs, osege, 1.0 / 3)
        for q in range(4):
            assert np.array_equal(planes[q], v["sbc"][:, :, 10 + q])


def test_oracle_matches_golden_fixture():
    """The pinning without the compiled reference at hand (GPU box): the fixture holds the reference's own outputs
    (tests/golden/make_golden.py clinic_fixture)."""
    gold = np.load(ROOT / "tests" / "golden" / "clinic_m2_14x14x6.npz")
    oc, mom, eos, flt = _setup(14, 14, 6)
    rho, M, up, zu = _oracle_step(oc, mom, eos)
    assert np.array_equal(rho, gold["rho"])
    assert np.array_equal(M.a["grad_p"], gold["grad_p"])
    assert np.array_equal(zu, gold["zu"]) and np.array_equal(up, gold["u_taup1"])
    _, _, upf, _ = _oracle_step(oc, mom, eos, flt)
    assert np.array_equal(upf, gold["u_taup1_filtered"])
    assert not np.array_equal(upf, up)


@pytest.mark.parametrize("imt,jmt,km", GRIDS)
def test_hostemu_equals_oracle(imt, jmt, km):
    import emu
    oc, mom, eos, flt = _setup(imt, jmt, km)
    rho, M, up, zu = _oracle_step(oc, mom, eos)
    E = emu.EmuMomentum(oc, mom, eos)
    assert np.array_equal(E.state()[:, :, 1:], rho[:, :, 1:])
    got_u, got_zu = E.clinic()
    assert np.array_equal(E.a["grad_p"], M.a["grad_p"])
    assert np.array_equal(got_zu, zu) and np.array_equal(got_u, up)
    _, _, upf, _ = _oracle_step(oc, mom, eos, flt)
    assert np.array_equal(E.filuv(oc.grid, flt), upf) and not np.array_equal(upf, up)
    psi = _psi(oc.grid, 1)
    assert np.array_equal(E.add_ext_mode(psi, oc.u), M.add_ext_mode(psi, oc.u))
    # the surface-velocity accumulators, all four phases of a coupling segment
    rng = np.random.default_rng(1)
    planes = [np.asfortranarray(rng.standard_normal((imt, jmt))) for _ in range(4)]
    for q, n in enumerate(("sbc_gu", "sbc_gv", "sbc_su", "sbc_sv")):
        E.a[n][...] = planes[q]
    for osegs, osege in ((1, 0), (0, 0), (0, 1), (1, 1)):
        M.sbcu("i", planes[0], planes[1], osegs, osege, 0.25)
        M.sbcu("a", planes[2], planes[3], osegs, osege, 0.25)
        E.sbcu(osegs | (osege << 1), 0.25)
        for q, n in enumerate(("sbc_gu", "sbc_gv", "sbc_su", "sbc_sv")):
            assert np.array_equal(E.a[n], planes[q])


def _gpu_model(oc, mom, eos):
    from uvic29_amd.tracer import TracerModel
    g = oc.grid
    m = TracerModel(g.imt, g.jmt, g.km, oc.cfg.nt, oc.cfg.nsrc, oc.cfg.ntnpzd, device=0)
    m.load_ocean(oc, *eos)
    m.load_momentum(oc, mom)
    return m


@pytest.mark.gpu
@pytest.mark.parametrize("imt,jmt,km", GRIDS + [(23, 17, 6), (38, 101, 19)])       # (the last two: not square, odd)
def test_gpu_state_and_clinic_equal_oracle(imt, jmt, km):
    oc, mom, eos, flt = _setup(imt, jmt, km)
    rho, M, up, zu = _oracle_step(oc, mom, eos)
    m = _gpu_model(oc, mom, eos)
    m.state()
    assert np.array_equal(m.download("rho")[:, :, 1:], rho[:, :, 1:])
    got_u, got_zu = m.clinic()
    assert np.array_equal(m.download("grad_p"), M.a["grad_p"])
    assert np.array_equal(got_zu, zu)
    assert np.array_equal(got_u, up)
    # with the polar filter
    _, _, upf, _ = _oracle_step(oc, mom, eos, flt)
    m.set_filter_u(oc, flt)
    got_u, got_zu = m.clinic()
    assert np.array_equal(got_zu, zu) and np.array_equal(got_u, upf)
    m.close()


@pytest.mark.gpu
def test_gpu_clinic_on_the_refined_grid_and_in_two_slabs():
    """BASELINE config 5's grid (202x202x32): state + clinic bit-identical to the oracle, computed whole and as two
    latitude slabs (uvic_gpu_set_shard js..je: each call fills its own U rows, the inputs around them are present)."""
    oc, mom, eos, flt = _setup(202, 202, 32)
    rho, M, up, zu = _oracle_step(oc, mom, eos, flt)
    m = _gpu_model(oc, mom, eos)
    m.set_filter_u(oc, flt)
    m.state()
    got_u, got_zu = m.clinic()
    assert np.array_equal(got_zu, zu) and np.array_equal(got_u, up)
    for n in ("up1", "up2"):
        m.upload(n, np.zeros(m.shape(n), order="F"))
    m.upload("zu", np.zeros(m.shape("zu"), order="F"))
    for js, je in ((2, 101), (102, 201)):
        m.set_shard(js=js, je=je)
        m.clinic_only()
    m.set_shard(js=2, je=201)
    got_u = np.stack([m.download("up1"), m.download("up2")], axis=-1)
    assert np.array_equal(m.download("zu"), zu) and np.array_equal(got_u, up)
    m.close()


@pytest.mark.gpu
def test_gpu_clinic_matches_golden_fixture():
    gold = np.load(ROOT / "tests" / "golden" / "clinic_m2_14x14x6.npz")
    oc, mom, eos, flt = _setup(14, 14, 6)
    m = _gpu_model(oc, mom, eos)
    m.state()
    got_u, got_zu = m.clinic()
    assert np.array_equal(got_zu, gold["zu"]) and np.array_equal(got_u, gold["u_taup1"])
    m.set_filter_u(oc, flt)
    got_u, _ = m.clinic()
    assert np.array_equal(got_u, gold["u_taup1_filtered"])
    m.close()


@pytest.mark.gpu
def test_gpu_sbc_accumulation_equals_oracle():
    oc, mom, eos, _ = _setup(14, 14, 6)
    rho, M, up, zu = _oracle_step(oc, mom, eos)
    m = _gpu_model(oc, mom, eos)
    m.state()
    rng = np.random.default_rng(1)
    names = ("sbc_gu", "sbc_gv", "sbc_su", "sbc_sv")
    planes = [np.asfortranarray(rng.standard_normal((14, 14))) for _ in range(4)]
    for q, n in enumerate(names):
        m.upload(n, planes[q])
    for osegs, osege in ((1, 0), (0, 0), (0, 1), (1, 1)):
        M.sbcu("i", planes[0], planes[1], osegs, osege, 0.25)
        M.sbcu("a", planes[2], planes[3], osegs, osege, 0.25)
        m.clinic(accumulate_sbc=True, osegs=bool(osegs), osege=bool(osege), rts=0.25)
        for q, n in enumerate(names):
            assert np.array_equal(m.download(n), planes[q])
    m.close()


@pytest.mark.gpu
def test_gpu_clinic_properties_at_full_size():
    """Size-independent properties on BASELINE's grid: internal modes have no vertical mean, land stays at rest,
    the cyclic columns are images, zu is the depth average of the tendency."""
    oc, mom, eos, flt = _setup(102, 102, 19)
    g, topo = oc.grid, oc.topo
    m = _gpu_model(oc, mom, eos)
    m.set_filter_u(oc, flt)
    m.state()
    u, zu = m.clinic()
    for n in range(2):
        mean = np.einsum("ikj,k->ij", u[..., n], g.dzt) * mom.hr
        scale = np.abs(u[..., n]).max()
        assert np.abs(mean[1:-1, 1:-1]).max() <= 1e-12 * scale
        assert np.all(u[..., n][topo.umask == 0.0][...] == 0.0) or np.all((u[..., n] * (1 - topo.umask))[1:-1, :, 1:-1] == 0.0)
        assert np.array_equal(u[0, :, 1:-1, n], u[-2, :, 1:-1, n]) and np.array_equal(u[-1, :, 1:-1, n], u[1, :, 1:-1, n])
    assert np.all(zu[topo.kmu == 0] == 0.0)
    m.close()


# ---- the whole memory-window loop on the device (mom.F:289-408 without tropic) -------------------------------------------
def _psi(g, n):
    """A smooth stream function that changes from step to step (stands in for tropic's solution), zero on the rim."""
    lam = 2.0 * np.pi * (np.arange(1, g.imt + 1) - 1.0) / (g.imt - 2)
    p = 2.0e11 * np.sin(g.phi * 2.0)[None, :] * (1.0 + 0.3 * np.sin(lam + 0.4 * n))[:, None] * np.cos(0.3 * n)
    p = np.asfortranarray(p)
    p[0] = p[-2]
    p[-1] = p[1]
    return p


def _reference_loop(oc, mom, nsteps, filters, shim=False, on_host=(), segment=0, tavg=(), tidal=None, tsi=False, gyre=(), glen=(), forward=(), nmix=None, vary=False, perturb=(), clock=False):
    """mom.F's loop through the compiled reference's own routines (oracle/_ref build "m2"): add_ext_mode, state, adv_vel,
    isopyc, "+K33", setvbc, tracer, clinic; the time levels of t and u rotated as the memory window does.
    shim: the same loop with `tracer` and `clinic` replaced by the package's Fortran overlays (the host-side routines still
    run, as in the model, on whatever the host arrays hold).  on_host: steps with a diagnostic switch set (term balances),
    which the overlays hand to the reference routines.  tavg: time-average steps (timavgperts), which stay on the device;
    what `diag -> avgvar` reads of the memory window after them (t(tau), u(tau), adv_vbt) is appended to the returned list.
    segment: ocean steps per coupling segment (isbcu/asbcu sums of clinic).  tidal: the tidal dissipation fields -- the
    reference's own `vmixc` (tidal mixing + K33) then follows `isopyc` instead of the "+K33" done here; the host's K33
    after the last step is appended to what a shim run returns (stale where the overlays left isopyc to the device).
    tsi: every step a time-step-monitor step (tsiperts, as with the shipped run/control.in); the integrals of every step
    (tbar, travar, dtabs, ektot) are returned under "tsi" beside the host's K33.  gyre: steps with gyrets set, which only the
    `tracer` overlay hands to the reference routine (`clinic` stays on the device -- or, on step 1, finds no device instance
    yet and goes to the reference routine as well).  glen: steps with glents set, which only the `clinic` overlay hands to
    the reference routine.  forward: forward ("mixing") steps -- c2dt = dt and, as loadmw does for the wide-open window
    (loadmw.F:99-102), the index taum1 aliased to tau for the step; nmix: what the overlay predicts the next step's kind
    from (switch.F:217-223: a mixing step when mod(itt, nmix) = 1).  vary: the surface heat and salt fluxes and the wind
    stress (sbc planes, which setvbc turns into stf and smf) differ from step to step -- the device copies of the step's
    inputs are taken in turn, a stale one would show.  perturb: after these steps the HOST changes t and u (as a restart
    read or a nudging term would); a shim run brings its copies up to date first (clinic_gpu_flush, tracer_gpu_flush) and
    says so afterwards (tracer_gpu_invalidate).  clock: relyr advances every step (the overlay extrapolates it for the
    sources it computes ahead and accepts its guess when it is right to rounding: runs that are compared bit for bit between
    two modes of the overlay keep the clock still)."""
    import refdriver
    g = oc.grid
    R = refdriver.RefOcean(oc, shim=shim)
    R.set_momentum(mom)
    if filters:
        from uvic29_amd import synthetic
        R.set_filter(synthetic.make_filter(g, g.km))
        R.set_filter_u(synthetic.make_filter_u(g, g.km))
    S, v = R.ref.set, R.v
    # setvbc takes the surface tracer fluxes from sbc: give it the synthetic stf there (btf is zero without O_gthflx)
    np_ = v["sbc"].shape[2]
    S("ihflx", np_ - 3); S("isflx", np_ - 2)
    v["sbc"][:, :, np_ - 4] = oc.stf[:, :, 0]
    v["sbc"][:, :, np_ - 3] = oc.stf[:, :, 1]
    if segment:
        S("igu", np_ - 9); S("igv", np_ - 8); S("isu", np_ - 7); S("isv", np_ - 6); S("ntspos", segment)
        v["sbc"][:, :, np_ - 10:np_ - 6] = np.random.default_rng(3).standard_normal((g.imt, g.jmt, 4))
    R.set_step_kind(False)          # leapfrog steps throughout (switch.h)
    v["u"][..., 2] = 0.0            # (the COMMON blocks outlive a model instance in this process)
    if tidal is not None:           # COMMON /tdr/ (tidal_kv.h) and what vmixc finds below the bottom level on the first step
        for name in ("edrm2", "edrs2", "edrk1", "edro1"):
            v[name][...] = getattr(tidal, name)
        for name in ("zetar", "ogamma", "gravrho0r", "kappa_h"):
            S(name, getattr(tidal, name))
        v["diff_cbt"][...] = oc.diff_cbt_bg[:, :, 1:g.jmt - 1]
        v["k33"][...] = 0.0
    tsis = []
    S("tsiperts", 1 if tsi else 0)
    if "relyr" in v:                # MOBI: the overlay predicts the next step's light from these (switch.F:217-223, tmngr.F:330-367)
        S("nmix", 0); S("prelyr", float(v["relyr"][0]))
    if nmix is not None:
        S("nmix", nmix)
    zus = []
    for n in range(1, nsteps + 1):
        S("itt", n)
        if segment:
            S("osegs", 1 if (n - 1) % segment == 0 else 0); S("osege", 1 if n % segment == 0 else 0)
        # (every switch on every step: the COMMON blocks outlive the model instances of a process)
        S("trmbts", 1 if n in on_host else 0)
        S("timavgperts", 1 if n in tavg else 0)
        S("gyrets", 1 if n in gyre else 0)
        S("glents", 1 if n in glen else 0)
        fwd = n in forward
        R.set_step_kind(fwd)
        S("c2dtuv", mom.dtuv if fwd else 2.0 * mom.dtuv)
        S("taum1", 0 if fwd else -1)          # (mw.h: the time levels are indexed -1:1)
        if vary and segment and "dnswr" in v and (n - 1) % segment == 0:
            # ... and at a segment's first step the atmosphere and ice fields MOBI reads (the coupler runs between segments)
            kseg = (n - 1) // segment
            fo = oc.forcing
            v["dnswr"][...] = fo.dnswr * (1.0 + 0.1 * (kseg % 3)) * (1.0 + 0.1 * np.sin(0.3 * np.arange(g.imt) + kseg))[:, None]
            v["aice"][:, :, 1] = np.clip(fo.aice + 0.2 * (kseg % 3) * (np.cos(0.2 * np.arange(g.jmt)) > 0.5)[None, :], 0.0, 1.0)
            v["hice"][:, :, 1] = fo.hice + 10.0 * (kseg % 3) * (v["aice"][:, :, 1] > 0)
            S("co2ccn", fo.co2ccn + 10.0 * kseg)
        if clock and "relyr" in v:      # the model's clock advances (tmngr.F:330-367), here across a month boundary
            dyr = oc.params.dtts / (365.0 * 86400.0)
            S("prelyr", (1.0 / 12.0 + (n - 5.3) * dyr) % 1.0)
            v["relyr"][...] = (1.0 / 12.0 + (n - 4.3) * dyr) % 1.0
        if vary:
            w = 1.0 + 0.2 * np.sin(1.7 * n)
            v["sbc"][:, :, np_ - 4] = oc.stf[:, :, 0] * w
            v["sbc"][:, :, np_ - 3] = oc.stf[:, :, 1] * (2.0 - w)
            v["sbc"][:, :, int(R.ref.get("itaux")) - 1] = mom.smf[..., 0] * (2.0 - w)
            v["sbc"][:, :, int(R.ref.get("itauy")) - 1] = mom.smf[..., 1] * w
        if tsi:                     # diagi zeroes them at the start of every step (source/mom/diagi.F:193-205)
            for name in ("tbar", "travar", "dtabs", "ektot"):
                v[name][...] = 0.0
        R.add_ext_mode(_psi(g, n), "tau")
        if n == 1:
            R.add_ext_mode(_psi(g, 0), "tau-1")
        R.state()
        R.ref.call("adv_vel", 0, 1, g.jmt, 2, g.imt - 1)
        R.isopyc()
        if tidal is not None:
            R.ref.call("vmixc", 0, 1, g.jmt, 2, g.imt - 1)
        else:
            R.add_k33()
        R.setvbc()
        R.tracer()
        _, zu, _ = R.clinic()
        zus.append(zu)
        if tsi:
            tsis.append({name: np.array(v[name], order="F") for name in ("tbar", "travar", "dtabs", "ektot")})
        if segment and n % segment == 0:
            zus.append(np.array(v["sbc"][:, :, np_ - 10:np_ - 6], order="F"))     # the averages the atmosphere reads
        if n in forward:
            S("taum1", -1)
        if n in tavg:      # what avgvar (diag.F:138-147) reads next
            zus.append(np.array(v["t"][:, :, 1:-1, :, 1], order="F"))
            zus.append(np.array(v["u"][:, :, 1:-1, :, 1], order="F"))
            zus.append(np.array(v["adv_vbt"][:, :, :-1], order="F"))
        resident = shim and hasattr(R.ref.lib, "tracer_gpu_invalidate_")
        if n in perturb and resident:
            R.ref.call("clinic_gpu_flush")          # u(tau+1), u(tau), u(tau-1) of this step into their slots
        R.rotate()
        u = v["u"]
        u[..., 0] = u[..., 1]
        u[..., 1] = u[..., 2]
        if n in perturb:
            if resident:
                R.ref.call("tracer_gpu_flush")      # t(tau-1), t(tau) of the coming step
            b1 = 1.0 + 1e-3 * np.sin(np.arange(g.imt))
            b1[0], b1[-1] = b1[-2], b1[1]           # (the cyclic image columns stay images)
            bump = b1[:, None, None, None]
            v["t"][..., 1] *= bump
            v["u"][..., 1] *= bump
            if resident:
                R.ref.call("tracer_gpu_invalidate")
    if shim and hasattr(R.ref.lib, "clinic_gpu_flush_"):
        # resident overlays: what the device holds for the coming step, into the host's (already rotated) slots
        stale = np.array(v["u"][..., 1], order="F")
        R.ref.call("tracer_gpu_flush")
        v["u"][..., 2] = 0.0
        R.ref.call("clinic_gpu_flush")      # u(tau+1), u(tau), u(tau-1) of the last step
        u = v["u"]
        if np.any(u[..., 2] != 0.0):
            last_tau, last_taup1 = np.array(u[..., 1], order="F"), np.array(u[..., 2], order="F")
        else:      # the last step left u with the host (a forward step, a step of the reference routines): nothing to bring down
            last_tau, last_taup1 = np.array(u[..., 0], order="F"), np.array(u[..., 1], order="F")
        if tidal is not None:
            return (np.array(v["t"][..., 1], order="F"), last_taup1, last_tau, zus, stale,
                    {"k33": np.array(v["k33"], order="F"), "adv_vnt": np.array(v["adv_vnt"], order="F"), "tsi": tsis,
                     "rho": np.array(v["rho"], order="F")})
        return np.array(v["t"][..., 1], order="F"), last_taup1, last_tau, zus, stale
    if tidal is not None:
        return (np.array(v["t"][..., 1], order="F"), np.array(v["u"][..., 1], order="F"), np.array(v["u"][..., 0], order="F"), zus,
                {"k33": np.array(v["k33"], order="F"), "adv_vnt": np.array(v["adv_vnt"], order="F"), "tsi": tsis,
                 "rho": np.array(v["rho"], order="F")})
    return np.array(v["t"][..., 1], order="F"), np.array(v["u"][..., 1], order="F"), np.array(v["u"][..., 0], order="F"), zus


@pytest.mark.parametrize("imt,jmt,km,nsteps", [(14, 14, 6, 6), (102, 102, 19, 3)])
def test_reference_loop_is_reproducible(imt, jmt, km, nsteps):
    """The driver of the loop test itself: two runs of the reference loop agree (guards the fixture below)."""
    import refmodel
    if not refmodel.available("m2", imt, jmt, km):
        pytest.skip("oracle/_ref build m2 not present")
    oc, mom, _, _ = _setup(imt, jmt, km)
    a = _reference_loop(oc, mom, 2, True)
    b = _reference_loop(oc, mom, 2, True)
    assert all(np.array_equal(x, y) for x, y in zip(a[:3], b[:3]))
    assert np.abs(a[1]).max() > 0.1


@pytest.mark.gpu
@pytest.mark.parametrize("imt,jmt,km,nsteps", [(14, 14, 6, 24), (102, 102, 19, 8)])
def test_gpu_memory_window_loop_equals_reference(imt, jmt, km, nsteps):
    """Everything of mom.F's loop but `tropic` on the device, several steps, tracers and velocities resident: T, S, u at
    both time levels and zu of every step bit-identical to the reference's own routines (exact transport kernels: T and
    S are pure transport; the momentum kernels have one formulation).  (24 steps on the small grid: the synthetic
    forcing with the tracers' accelerated time step lets the velocities grow, beyond ~30 steps the fields are no
    longer a meaningful ocean.)"""
    import refmodel
    from uvic29_amd import synthetic
    from uvic29_amd.tracer import OceanLoop
    if not refmodel.available("m2", imt, jmt, km):
        pytest.skip("oracle/_ref build m2 did not travel with the tree")
    oc, mom, eos, flt_u = _setup(imt, jmt, km)
    t_ref, u_ref, um_ref, zus = _reference_loop(oc, mom, nsteps, True)
    oc.btf[...] = 0.0
    m = _gpu_model(oc, mom, eos)
    m.set_exact(True)
    m.load_velocity(oc)
    m.set_filter(oc, synthetic.make_filter(oc.grid, km))
    m.set_filter_u(oc, flt_u)
    loop = OceanLoop(m, oc.params.dtts, mom.dtuv)
    for n in range(1, nsteps + 1):
        zu = loop.step(_psi(oc.grid, n), _psi(oc.grid, 0) if n == 1 else None)
        assert np.array_equal(zu, zus[n - 1]), n
    t = m.download("t_tau")
    assert np.array_equal(t[:, :, 1:-1], t_ref[:, :, 1:-1])
    u = np.stack([m.download("u1"), m.download("u2")], axis=-1)
    um = np.stack([m.download("um1"), m.download("um2")], axis=-1)
    assert np.array_equal(u[:, :, 1:-1], u_ref[:, :, 1:-1]) and np.array_equal(um[:, :, 1:-1], um_ref[:, :, 1:-1])
    m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("exact", [True, False])
@pytest.mark.parametrize("imt,jmt,km,nsteps", [(14, 14, 6, 9), (102, 102, 19, 4)])
def test_fortran_overlays_keep_the_velocities_on_the_device(imt, jmt, km, nsteps, exact, monkeypatch):
    """The same loop through the Fortran boundary (tracer_gpu.F + clinic_gpu.F, UVIC_RESIDENT=2): after the first `clinic`
    on the device u stays there -- rotation of the levels, add_ext_mode and adv_vel happen on the device at the start of
    the next step, per step psi and the wind stress go up and zu comes back -- while the host-side routines of the loop
    (loadmw's add_ext_mode, state, adv_vel, setvbc) keep running on the host's stale copy as they would in the model.
    zu of every step, the isbcu/asbcu averages at the end of every segment, and T, S, u after the last step equal the
    reference's own loop bit for bit; one step in the middle carries a diagnostic switch and goes through the reference
    routines (u comes down, adv_vel and setvbc are redone on the host, u(tau+1) goes back up); one is a time-average step,
    which stays on the device and leaves t(tau), u(tau) and adv_vbt on the host for avgvar."""
    import refmodel
    if not (refmodel.available("m2", imt, jmt, km) and refmodel.available("m2", imt, jmt, km, shim=True)):
        pytest.skip("oracle/_ref build m2 (reference and shim) did not travel with the tree")
    monkeypatch.setenv("UVIC_RESIDENT", "2")
    if exact:
        monkeypatch.setenv("UVIC_EXACT", "1")
    else:       # the production default: T and S still go through the bit-exact kernels
        monkeypatch.delenv("UVIC_EXACT", raising=False)
    oc, mom, _, _ = _setup(imt, jmt, km)
    on_host = (nsteps - 3,) if nsteps > 5 else ()
    tavg = (nsteps - 1,) if nsteps > 5 else (3,)
    t_ref, u_ref, um_ref, zus = _reference_loop(oc, mom, nsteps, True, on_host=on_host, segment=3, tavg=tavg)
    out = _reference_loop(oc, mom, nsteps, True, shim=True, on_host=on_host, segment=3, tavg=tavg)
    if len(out) != 5:
        pytest.skip("oracle/_ref shim predates the resident velocities")
    t, u, um, got, stale = out
    assert len(got) == len(zus)
    for n, (a, b) in enumerate(zip(got, zus)):
        assert np.array_equal(a, b), n
    assert np.array_equal(t[:, :, 1:-1, :2], t_ref[:, :, 1:-1, :2])
    assert np.array_equal(u[:, :, 1:-1], u_ref[:, :, 1:-1]) and np.array_equal(um[:, :, 1:-1], um_ref[:, :, 1:-1])
    assert not np.array_equal(stale[:, :, 1:-1], u_ref[:, :, 1:-1])     # resident for real: the host's copy was stale


@pytest.mark.gpu
@pytest.mark.parametrize("exact", [True, False])
@pytest.mark.parametrize("imt,jmt,km,nsteps", [(14, 14, 6, 9), (102, 102, 19, 5)])
def test_fortran_overlays_leave_isopyc_and_vmixc_to_the_device(imt, jmt, km, nsteps, exact, monkeypatch):
    """UVIC_RESIDENT=3 (tracer_gpu.F + clinic_gpu.F + mixing_gpu.F): the host's `isopyc` and `vmixc` -- most of what the
    host still does per step once `tracer` and `clinic` are served by the device -- `adv_vel` and loadmw's `state` are left
    out on the steps the `tracer` overlay takes; no diff_cbt goes up, the device forms it as vmixc.F does (tidal mixing from the stratification + K33).
    Against the reference's own loop with its own isopyc and vmixc: zu of every step, the segment averages, T, S and u
    bit for bit; one step carries a diagnostic switch (both host routines run again, `tracer_cpu` takes their products),
    one is a time-average step (they run for isopyc's own averages, the step stays on the device)."""
    import refmodel
    from uvic29_amd import synthetic
    if not (refmodel.available("m2", imt, jmt, km) and refmodel.available("m2", imt, jmt, km, shim=True)):
        pytest.skip("oracle/_ref build m2 (reference and shim) did not travel with the tree")
    monkeypatch.setenv("UVIC_RESIDENT", "3")
    if exact:
        monkeypatch.setenv("UVIC_EXACT", "1")
    else:
        monkeypatch.delenv("UVIC_EXACT", raising=False)
    oc, mom, _, _ = _setup(imt, jmt, km)
    tid = synthetic.make_tidal(oc.grid, oc.topo, oc.params.kappa_h)
    on_host = (nsteps - 4,) if nsteps > 5 else ()
    tavg = (nsteps - 2,) if nsteps > 5 else (3,)
    t_ref, u_ref, um_ref, zus, host_ref = _reference_loop(oc, mom, nsteps, True, on_host=on_host, segment=3, tavg=tavg, tidal=tid)
    out = _reference_loop(oc, mom, nsteps, True, shim=True, on_host=on_host, segment=3, tavg=tavg, tidal=tid)
    if len(out) != 6:
        pytest.skip("oracle/_ref shim predates mixing_gpu.F")
    t, u, um, got, stale, host = out
    assert len(got) == len(zus)
    for n, (a, b) in enumerate(zip(got, zus)):
        assert np.array_equal(a, b), n
    assert np.array_equal(t[:, :, 1:-1, :2], t_ref[:, :, 1:-1, :2])
    assert np.array_equal(u[:, :, 1:-1], u_ref[:, :, 1:-1]) and np.array_equal(um[:, :, 1:-1], um_ref[:, :, 1:-1])
    # left out for real: the host's K33, adv_vnt and rho are not the last step's
    assert not np.array_equal(host["k33"], host_ref["k33"]) and not np.array_equal(host["adv_vnt"], host_ref["adv_vnt"])
    assert not np.array_equal(host["rho"], host_ref["rho"])


@pytest.mark.gpu
@pytest.mark.parametrize("level", ["2", "3"])
@pytest.mark.parametrize("exact", [False, True])
def test_fortran_overlays_when_the_run_opens_on_the_host(exact, level, monkeypatch):
    """A run whose FIRST step the `tracer` overlay hands to the reference routine (here: gyrets; an Euler backward start or
    tavgts do the same): no device instance exists when `clinic` is called, which then takes the reference routine too
    instead of stopping; the device comes in at step 2.  Another such step in mid-run: `tracer_cpu` with the tracers brought
    down, `clinic` on the device (which then takes the advective velocities from the host: nobody formed them on the
    device).  And the converse, a step only `clinic` hands over (glents): `tracer` on the device, u brought down for
    `clinic_cpu`, u(tau+1) back up.  With UVIC_RESIDENT=3 the host's isopyc, vmixc, adv_vel, state run on exactly the steps
    that need them.  Everything bit for bit against the reference's own loop."""
    import refmodel
    from uvic29_amd import synthetic
    imt, jmt, km, nsteps = 14, 14, 6, 8
    if not (refmodel.available("m2", imt, jmt, km) and refmodel.available("m2", imt, jmt, km, shim=True)):
        pytest.skip("oracle/_ref build m2 (reference and shim) did not travel with the tree")
    monkeypatch.setenv("UVIC_RESIDENT", level)
    if exact:
        monkeypatch.setenv("UVIC_EXACT", "1")
    else:
        monkeypatch.delenv("UVIC_EXACT", raising=False)
    oc, mom, _, _ = _setup(imt, jmt, km)
    tid = synthetic.make_tidal(oc.grid, oc.topo, oc.params.kappa_h)
    t_ref, u_ref, um_ref, zus, _ = _reference_loop(oc, mom, nsteps, True, segment=3, tidal=tid, gyre=(1, 4), glen=(6,), vary=True)
    out = _reference_loop(oc, mom, nsteps, True, shim=True, segment=3, tidal=tid, gyre=(1, 4), glen=(6,), vary=True)
    if len(out) != 6:
        pytest.skip("oracle/_ref shim predates mixing_gpu.F")
    t, u, um, got, _, _ = out
    for n, (a, b) in enumerate(zip(got, zus)):
        assert np.array_equal(a, b), n
    assert np.array_equal(t[:, :, 1:-1], t_ref[:, :, 1:-1])
    assert np.array_equal(u[:, :, 1:-1], u_ref[:, :, 1:-1]) and np.array_equal(um[:, :, 1:-1], um_ref[:, :, 1:-1])


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", ["m2", "t30"])
@pytest.mark.parametrize("level", ["1", "2", "3"])
def test_fortran_overlays_when_the_host_changes_the_state(level, cfg, monkeypatch):
    """In mid-run the host changes t and u between two steps (a restart read, nudging, assimilation).  With the state
    resident the caller brings the host copies up to date first (clinic_gpu_flush, tracer_gpu_flush), makes its change and
    says so (tracer_gpu_invalidate): the next step takes its time levels from the host's arrays and drops what the device
    had computed ahead from its own.  Against the reference's loop with the same change; production arithmetic."""
    import refmodel
    from uvic29_amd import synthetic
    imt, jmt, km, nsteps = 14, 14, 6, 8
    if not (refmodel.available(cfg, imt, jmt, km) and refmodel.available(cfg, imt, jmt, km, shim=True)):
        pytest.skip("oracle/_ref build %s (reference and shim) did not travel with the tree" % cfg)
    monkeypatch.setenv("UVIC_RESIDENT", level)
    monkeypatch.delenv("UVIC_EXACT", raising=False)
    oc = synthetic.make_ocean(cfg, imt, jmt, km)
    mom = synthetic.make_momentum(oc.grid, oc.topo, oc.u, anisotropic=True)
    tid = synthetic.make_tidal(oc.grid, oc.topo, oc.params.kappa_h)
    kw = dict(segment=3, tidal=tid, tsi=(cfg == "t30"), vary=True, perturb=(3, 5))
    t_ref, u_ref, um_ref, zus, _ = _reference_loop(oc, mom, nsteps, True, **kw)
    out = _reference_loop(oc, mom, nsteps, True, shim=True, **kw)
    if len(out) != 6 or not hasattr(__import__("refdriver").RefOcean(oc, shim=True).ref.lib, "tracer_gpu_invalidate_"):
        pytest.skip("oracle/_ref shim predates tracer_gpu_invalidate")
    t, u, um, got, _, _ = out
    for n, (a, b) in enumerate(zip(got, zus)):
        assert np.array_equal(a, b), n
    assert np.array_equal(t[:, :, 1:-1, :2], t_ref[:, :, 1:-1, :2])
    for n, name in enumerate(oc.cfg.tracers):
        a, b = t[:, :, 1:-1, n], t_ref[:, :, 1:-1, n]
        assert np.abs(a - b).max() <= 1e-11 * np.abs(b).max(), (name, np.abs(a - b).max())
    assert np.array_equal(u[:, :, 1:-1], u_ref[:, :, 1:-1]) and np.array_equal(um[:, :, 1:-1], um_ref[:, :, 1:-1])


def _compare_runs(oc, ref_out, shim_out, level, tol=1e-11):
    """zu, segment averages and what a time-average step leaves on the host: bit for bit (t of the tracers other than T, S
    to `tol` of its maximum); t, u after the last step likewise."""
    t_ref, u_ref, um_ref, zus, _ = ref_out
    t, u, um, got, _, _ = shim_out
    assert len(got) == len(zus)
    for n, (a, b) in enumerate(zip(got, zus)):
        if a.ndim == 4 and a.shape[-1] == oc.cfg.nt and oc.cfg.nt > 2:
            assert np.array_equal(a[..., :2], b[..., :2]), n
            scale = np.abs(b).max(axis=(0, 1, 2), keepdims=True)
            assert (np.abs(a - b) <= tol * scale).all(), n
        else:
            assert np.array_equal(a, b), n
    assert np.array_equal(t[:, :, 1:-1, :2], t_ref[:, :, 1:-1, :2])
    for n, name in enumerate(oc.cfg.tracers):
        a, b = t[:, :, 1:-1, n], t_ref[:, :, 1:-1, n]
        assert np.abs(a - b).max() <= tol * np.abs(b).max(), (name, np.abs(a - b).max())
    assert np.array_equal(u[:, :, 1:-1], u_ref[:, :, 1:-1]), "u(tau+1) of the last step"
    assert np.array_equal(um[:, :, 1:-1], um_ref[:, :, 1:-1]), "u(tau) of the last step"


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(1, 1 + int(__import__("os").environ.get("UVIC_TEST_SCHEDULES", "8"))))
@pytest.mark.parametrize("cfg", ["m2", "t30"])
@pytest.mark.parametrize("level", ["0", "1", "2", "3"])
def test_fortran_overlays_random_schedules(level, cfg, seed, monkeypatch):
    """Twelve steps whose kinds are drawn at random (fixed seeds): forward steps, time-average steps, steps both overlays
    hand to the reference routines (trmbts), steps only `tracer` (gyrets) or only `clinic` (glents) hands over, and changes of
    t and u by the host in between -- in whatever combination the draw gives, from the first step on.  Production arithmetic,
    surface fluxes and wind changing every step, segments of three steps; against the reference's own loop."""
    import refmodel
    from uvic29_amd import synthetic
    imt, jmt, km, nsteps = 14, 14, 6, int(__import__("os").environ.get("UVIC_TEST_SCHEDULE_STEPS", "12"))
    if not (refmodel.available(cfg, imt, jmt, km) and refmodel.available(cfg, imt, jmt, km, shim=True)):
        pytest.skip("oracle/_ref build %s (reference and shim) did not travel with the tree" % cfg)
    if level == "0":      # the default: nothing resident, every call moves the state both ways
        monkeypatch.delenv("UVIC_RESIDENT", raising=False)
    else:
        monkeypatch.setenv("UVIC_RESIDENT", level)
    monkeypatch.delenv("UVIC_EXACT", raising=False)
    rng = np.random.default_rng(1000 * seed + 7)
    steps = np.arange(1, nsteps + 1)
    draw = lambda p: tuple(int(n) for n in steps[rng.random(nsteps) < p])
    kw = dict(segment=int(rng.integers(2, 5)), tsi=(cfg == "t30"), vary=True, clock=True, nmix=int(rng.integers(0, 5)),
              forward=draw(0.2), tavg=draw(0.25), on_host=draw(0.12), gyre=draw(0.12), glen=draw(0.12),
              perturb=tuple(n for n in draw(0.15) if n < nsteps))
    oc = synthetic.make_ocean(cfg, imt, jmt, km)
    mom = synthetic.make_momentum(oc.grid, oc.topo, oc.u, anisotropic=True)
    kw["tidal"] = synthetic.make_tidal(oc.grid, oc.topo, oc.params.kappa_h)
    ref_out = _reference_loop(oc, mom, nsteps, True, **kw)
    shim_out = _reference_loop(oc, mom, nsteps, True, shim=True, **kw)
    if len(shim_out) != 6:
        pytest.skip("oracle/_ref shim predates mixing_gpu.F")
    try:
        _compare_runs(oc, ref_out, shim_out, level)
    except AssertionError as e:
        raise AssertionError("schedule %r: %s" % ({k: v for k, v in kw.items() if k != "tidal"}, e))


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,nmix", [("m2", 0), ("m2", 3), ("t30", 0), ("t30", 3)])
@pytest.mark.parametrize("level", ["2", "3"])
@pytest.mark.parametrize("exact", [False, True])
def test_fortran_overlays_through_forward_steps(exact, level, cfg, nmix, monkeypatch):
    """Forward ("mixing") steps in mid-run -- every nmix-th step of the shipped run -- with everything resident: the `tracer`
    overlay has the device read t(tau) as t(tau-1) and brings u down (the host's adv_vel runs again on it), `clinic` takes
    the step from the host's arrays and u(tau+1) stays on the device for the leapfrog steps that follow.  nmix = 3: the
    overlay sees them coming (no look-ahead into them); nmix = 0: it does not, and drops what it computed ahead.  T and S
    only (m2) and option set C with the time-step monitor on every step (t30).  Against the reference's own loop: T, S, u,
    zu bit for bit, the other tracers to the production tolerance."""
    import refmodel
    from uvic29_amd import synthetic
    imt, jmt, km, nsteps = 14, 14, 6, 9
    if not (refmodel.available(cfg, imt, jmt, km) and refmodel.available(cfg, imt, jmt, km, shim=True)):
        pytest.skip("oracle/_ref build %s (reference and shim) did not travel with the tree" % cfg)
    monkeypatch.setenv("UVIC_RESIDENT", level)
    if exact:
        monkeypatch.setenv("UVIC_EXACT", "1")
    else:       # the production schedule: T,S on their own stream, look-ahead chains, relaxed ordering
        monkeypatch.delenv("UVIC_EXACT", raising=False)
    oc = synthetic.make_ocean(cfg, imt, jmt, km)
    mom = synthetic.make_momentum(oc.grid, oc.topo, oc.u, anisotropic=True)
    tid = synthetic.make_tidal(oc.grid, oc.topo, oc.params.kappa_h)
    kw = dict(segment=3, tidal=tid, forward=(4, 7), nmix=nmix, tsi=(cfg == "t30"), vary=True)
    t_ref, u_ref, um_ref, zus, _ = _reference_loop(oc, mom, nsteps, True, **kw)
    out = _reference_loop(oc, mom, nsteps, True, shim=True, **kw)
    if len(out) != 6:
        pytest.skip("oracle/_ref shim predates mixing_gpu.F")
    t, u, um, got, _, _ = out
    for n, (a, b) in enumerate(zip(got, zus)):
        assert np.array_equal(a, b), n
    assert np.array_equal(t[:, :, 1:-1, :2], t_ref[:, :, 1:-1, :2])
    for n, name in enumerate(oc.cfg.tracers):
        a, b = t[:, :, 1:-1, n], t_ref[:, :, 1:-1, n]
        assert np.abs(a - b).max() <= 1e-11 * np.abs(b).max(), (name, np.abs(a - b).max())
    assert np.array_equal(u[:, :, 1:-1], u_ref[:, :, 1:-1]) and np.array_equal(um[:, :, 1:-1], um_ref[:, :, 1:-1])


@pytest.mark.gpu
@pytest.mark.parametrize("exact", [True, False])
def test_fortran_overlays_with_the_shipped_options_and_switches(exact, monkeypatch):
    """The nearest thing to the shipped run this tree can drive: option set C built as run/mk.in builds it (oracle/_ref "t30":
    MOBI nt=30 + O_stream_function, O_anisotropic_viscosity, O_ice_evp, O_time_step_monitor, tidal mixing), the switches of
    run/control.in (tsiperts on every step, ocean segments, a forward step every nmix-th step, a time-average step), both
    polar filters on, everything resident and isopyc, vmixc, adv_vel, state left to the device (UVIC_RESIDENT=3) -- mom.F's
    loop through the three overlays against the reference's own loop: zu, u and the segment averages of every step, T and S, the kinetic-energy integral bit for bit; the MOBI tracers
    and their integrals to the production tolerance (their sources go through the device's exp/log)."""
    import refmodel
    from uvic29_amd import synthetic
    imt, jmt, km, nsteps = 14, 14, 6, 7
    if not (refmodel.available("t30", imt, jmt, km) and refmodel.available("t30", imt, jmt, km, shim=True)):
        pytest.skip("oracle/_ref build t30 (reference and shim) did not travel with the tree")
    monkeypatch.setenv("UVIC_RESIDENT", "3")
    if exact:
        monkeypatch.setenv("UVIC_EXACT", "1")
    else:
        monkeypatch.delenv("UVIC_EXACT", raising=False)
    oc = synthetic.make_ocean("t30", imt, jmt, km)
    mom = synthetic.make_momentum(oc.grid, oc.topo, oc.u, anisotropic=True)
    tid = synthetic.make_tidal(oc.grid, oc.topo, oc.params.kappa_h)
    kw = dict(segment=3, tidal=tid, tsi=True, vary=True, tavg=(4, 5), forward=(4,), nmix=3)      # (step 4: forward AND time-average)
    t_ref, u_ref, um_ref, zus, host_ref = _reference_loop(oc, mom, nsteps, True, **kw)
    out = _reference_loop(oc, mom, nsteps, True, shim=True, **kw)
    if len(out) != 6:
        pytest.skip("oracle/_ref shim predates mixing_gpu.F")
    t, u, um, got, stale, host = out
    tol = 1e-11
    assert len(got) == len(zus)
    for n, (a, b) in enumerate(zip(got, zus)):
        if a.ndim == 4 and a.shape[-1] == oc.cfg.nt:      # t(tau) of every tracer, left on the host for avgvar on the time-average step
            assert np.array_equal(a[..., :2], b[..., :2]), n
            scale = np.abs(b).max(axis=(0, 1, 2), keepdims=True)
            assert (np.abs(a - b) <= tol * scale).all(), n
        else:
            assert np.array_equal(a, b), n
    assert np.array_equal(t[:, :, 1:-1, :2], t_ref[:, :, 1:-1, :2])
    for n, name in enumerate(oc.cfg.tracers):
        a, b = t[:, :, 1:-1, n], t_ref[:, :, 1:-1, n]
        assert np.abs(a - b).max() <= tol * np.abs(b).max(), (name, np.abs(a - b).max())
    assert np.array_equal(u[:, :, 1:-1], u_ref[:, :, 1:-1]) and np.array_equal(um[:, :, 1:-1], um_ref[:, :, 1:-1])
    for step, (a, b) in enumerate(zip(host["tsi"], host_ref["tsi"])):
        assert np.abs(b["ektot"]).max() > 0 and np.array_equal(a["ektot"], b["ektot"]), step
        for name in ("tbar", "travar", "dtabs"):
            assert np.abs(b[name]).max() > 0
            assert np.array_equal(a[name][:, :2], b[name][:, :2]), (step, name)          # T and S
            scale = np.abs(b[name]).max(axis=(0, 2), keepdims=True)
            assert (np.abs(a[name] - b[name]) <= tol * np.maximum(scale, 1e-300)).all(), (step, name)
    assert not np.array_equal(host["k33"], host_ref["k33"])


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(1, 1 + int(__import__("os").environ.get("UVIC_TEST_FULL_SCHEDULES", "2"))))
def test_full_grid_random_schedule_resident_equals_not_resident(seed, monkeypatch):
    """102x102x19, option set C as run/mk.in builds it, time-step monitor on every step: a randomly drawn schedule of
    sixteen steps (forward, time-average, handed-over steps, a change of the state by the host, new atmosphere and ice
    fields every segment) with everything resident and asynchronous (UVIC_RESIDENT=3: look-ahead chains, relaxed ordering,
    copies beside the kernels -- at this size the kernels are long enough for a wrong dependency to show) against the same
    schedule with nothing resident (every call moves the state both ways and waits): every tracer, u, zu and the integrals
    bit for bit, since the arithmetic is the same.  (The non-resident mode against the reference itself: the small grids.)"""
    import refmodel
    from uvic29_amd import synthetic
    imt, jmt, km, nsteps = 102, 102, 19, 16
    if not refmodel.available("t30", imt, jmt, km, shim=True):
        pytest.skip("oracle/_ref shim t30 102x102x19 did not travel with the tree")
    monkeypatch.delenv("UVIC_EXACT", raising=False)
    rng = np.random.default_rng(4242 + seed)
    steps = np.arange(1, nsteps + 1)
    draw = lambda p: tuple(int(n) for n in steps[rng.random(nsteps) < p])
    kw = dict(segment=4, tsi=True, vary=True, nmix=int(rng.integers(0, 5)), forward=draw(0.2), tavg=draw(0.2), on_host=draw(0.1),
              gyre=draw(0.1), glen=draw(0.1), perturb=tuple(n for n in draw(0.1) if n < nsteps))
    oc = synthetic.make_ocean("t30", imt, jmt, km)
    mom = synthetic.make_momentum(oc.grid, oc.topo, oc.u, anisotropic=True)
    kw["tidal"] = synthetic.make_tidal(oc.grid, oc.topo, oc.params.kappa_h)
    runs = {}
    for level in ("0", "3"):
        if level == "0":
            monkeypatch.delenv("UVIC_RESIDENT", raising=False)
        else:
            monkeypatch.setenv("UVIC_RESIDENT", level)
        out = _reference_loop(oc, mom, nsteps, True, shim=True, **kw)
        if len(out) != 6:
            pytest.skip("oracle/_ref shim predates mixing_gpu.F")
        runs[level] = out
    (t0, u0, um0, zu0, _, h0), (t3, u3, um3, zu3, _, h3) = runs["0"], runs["3"]
    what = {k: v for k, v in kw.items() if k != "tidal"}
    assert np.isfinite(t3).all(), what
    assert len(zu0) == len(zu3)
    for n, (a, b) in enumerate(zip(zu0, zu3)):
        assert np.array_equal(a, b), (n, what)
    assert np.array_equal(t0, t3), what
    assert np.array_equal(u0, u3) and np.array_equal(um0, um3), what
    for a, b in zip(h0["tsi"], h3["tsi"]):
        assert all(np.array_equal(a[n], b[n]) for n in a), what


@pytest.mark.gpu
def test_full_grid_with_mixing_on_the_device_equals_mixing_on_the_host(monkeypatch):
    """102x102x19, option set C as run/mk.in builds it, shipped switches, production arithmetic, twelve steps through the three
    overlays: with isopyc, vmixc and adv_vel left to the device (UVIC_RESIDENT=3) EVERY tracer, u and zu come out bit for bit
    as with the host's own routines feeding the overlays (UVIC_RESIDENT=2) -- the device's diff_cbt and velocities are the
    host's, bit for bit, so nothing downstream may differ.  (Level 2 against the reference itself: the tests above.)"""
    import refmodel
    from uvic29_amd import synthetic
    imt, jmt, km, nsteps = 102, 102, 19, 12
    if not refmodel.available("t30", imt, jmt, km, shim=True):
        pytest.skip("oracle/_ref shim t30 102x102x19 did not travel with the tree")
    monkeypatch.delenv("UVIC_EXACT", raising=False)
    oc = synthetic.make_ocean("t30", imt, jmt, km)
    mom = synthetic.make_momentum(oc.grid, oc.topo, oc.u, anisotropic=True)
    tid = synthetic.make_tidal(oc.grid, oc.topo, oc.params.kappa_h)
    runs = {}
    for level in ("2", "3"):
        monkeypatch.setenv("UVIC_RESIDENT", level)
        out = _reference_loop(oc, mom, nsteps, True, shim=True, segment=4, tidal=tid, tsi=True, vary=True)
        if len(out) != 6:
            pytest.skip("oracle/_ref shim predates mixing_gpu.F")
        runs[level] = out
    (t2, u2, um2, zu2, _, h2), (t3, u3, um3, zu3, _, h3) = runs["2"], runs["3"]
    assert np.isfinite(t3).all() and np.abs(u3).max() > 0.1
    assert np.array_equal(t2, t3)
    assert np.array_equal(u2, u3) and np.array_equal(um2, um3)
    assert all(np.array_equal(a, b) for a, b in zip(zu2, zu3))
    for a, b in zip(h2["tsi"], h3["tsi"]):
        assert all(np.array_equal(a[n], b[n]) for n in a)
    assert not np.array_equal(h2["k33"], h3["k33"])      # (the host's K33: current with level 2, stale with level 3)


@pytest.mark.gpu
def test_gpu_momentum_entry_points_fail_loudly():
    """Misuse is refused with a message, not computed through (the error convention of the C ABI: non-zero status and
    uvic_gpu_last_error)."""
    from uvic29_amd.capi import UvicGpuError
    from uvic29_amd.tracer import TracerModel
    oc, mom, eos, flt = _setup(14, 14, 6)
    m = TracerModel(14, 14, 6, oc.cfg.nt, oc.cfg.nsrc, oc.cfg.ntnpzd, device=0)
    m.load_ocean(oc, *eos)
    with pytest.raises(UvicGpuError, match="set_clinic_params"):
        m.clinic_only()                                   # before the scalars of clinic were given
    with pytest.raises(UvicGpuError, match="level"):
        m.add_ext_mode(2)
    m.load_momentum(oc, mom)
    bad = type(flt)(**{**flt.__dict__, "jfu1": 9, "jfu2": 5})
    with pytest.raises(UvicGpuError, match="rows out of range"):
        m.set_filter_u(oc, bad)
    m.set_shard(n0=1, nt_local=1)                         # a tracer shard without T and S cannot form rho
    with pytest.raises(UvicGpuError, match="T and S"):
        m.state()
    m.close()
    from uvic29_amd.tmm import TmmOperator
    from uvic29_amd import synthetic
    c30 = synthetic.make_ocean("c30", 14, 14, 6)
    with pytest.raises(UvicGpuError, match="at least 4 columns"):
        TmmOperator(c30.cfg, c30.grid, 2)
    op = TmmOperator(c30.cfg, c30.grid, 8)
    with pytest.raises(UvicGpuError, match="tmm_set_mobi"):
        op.sources(np.zeros((8, 6, c30.cfg.nt)), 2.0 * c30.params.dtts)
    op.close()
