"""The drop-in boundary end to end: the reference model's own COMMON blocks and
call sequence (isopyc -> "+K33" -> tracer, source/mom/mom.F:340-389), with
`tracer` replaced by the package's Fortran overlay (uvic2.9_amd/fortran/
tracer_gpu.F -> ISO_C_BINDING -> libuvic_gpu.so -> HIP kernels), against the
unmodified reference.  Needs the libraries oracle/build_ref.py produces in the
build container (they travel to the GPU box as built artefacts)."""
import numpy as np
import pytest

from uvic29_amd import synthetic
import refmodel

pytestmark = pytest.mark.gpu


def _arith(monkeypatch, exact):
    """the overlay creates its own handle and reads the environment: UVIC_EXACT=1 = bit-exact arithmetic, unset = the
    production default a maintainer gets (column kernels, device MOBI shortcuts)"""
    if exact:
        monkeypatch.setenv("UVIC_EXACT", "1")
    else:
        monkeypatch.delenv("UVIC_EXACT", raising=False)


PROD_TOL = 1e-11     # production path through the overlay, relative to max|field| (one step 1e-13, MOBI sources 1e-11)


@pytest.mark.parametrize("exact", [True, False])
@pytest.mark.parametrize("cfg,dims", [("p2", (14, 14, 6)), ("c30", (14, 14, 6)), ("c30", (102, 102, 19)),
                                      ("f18", (14, 14, 6)), ("s37", (14, 14, 6))])   # (f18, s37: SURVEY.md §2c sets F and run/mk.in's)
def test_overlay_tracer_matches_reference_tracer(cfg, dims, exact, monkeypatch):
    _arith(monkeypatch, exact)
    if not (refmodel.available(cfg, *dims) and refmodel.available(cfg, *dims, shim=True)):
        pytest.skip("oracle/_ref reference/shim libraries did not travel with the tree")
    import refdriver
    oc = synthetic.make_ocean(cfg, *dims)
    ref = refdriver.RefOcean(oc)
    want = ref.step().copy()
    shim = refdriver.RefOcean(oc, shim=True)
    got = shim.step().copy()
    jmt = dims[1]
    # T and S: pure transport, bit-exact in the exact arithmetic
    if exact:
        assert np.array_equal(got[:, :, 1:jmt - 1, :2], want[:, :, 1:jmt - 1, :2])
    for n, name in enumerate(oc.cfg.tracers):
        a, b = got[:, :, 1:jmt - 1, n], want[:, :, 1:jmt - 1, n]
        assert np.abs(a - b).max() <= PROD_TOL * np.abs(b).max(), (name, np.abs(a - b).max())


def test_overlay_tracer_with_polar_filter_matches_reference(monkeypatch):
    """The same with the polar Fourier filter of `tracer` switched on (O_fourfil; filter rows from setcom.F's
    latitudes): the overlay hands jfrst, jft0-2 of index.h to uvic_gpu_set_filter once, the device then
    filters t(taup1) after convection as tracer.F:1245 does."""
    monkeypatch.setenv("UVIC_EXACT", "1")
    cfg, dims = "p2", (14, 14, 6)
    if not (refmodel.available(cfg, *dims) and refmodel.available(cfg, *dims, shim=True)):
        pytest.skip("oracle/_ref reference/shim libraries did not travel with the tree")
    import refdriver
    oc = synthetic.make_ocean(cfg, *dims)
    flt = synthetic.make_filter(oc.grid, dims[2])
    ref = refdriver.RefOcean(oc)
    if not hasattr(ref.ref.lib, "findex_"):
        pytest.skip("oracle/_ref predates findex")
    plain = ref.step().copy()
    ref = refdriver.RefOcean(oc)                 # (one library = one set of COMMON blocks: start over)
    ref.set_filter(flt)
    want = ref.step().copy()
    assert (want != plain).any()                 # the filter did something
    shim = refdriver.RefOcean(oc, shim=True)
    shim.set_filter(flt)
    got = shim.step().copy()
    assert np.array_equal(got[:, :, 1:-1], want[:, :, 1:-1])


def _same(a, b, exact, tol=PROD_TOL):
    if exact:
        return np.array_equal(a, b)
    return np.abs(a - b).max() <= tol * max(np.abs(b).max(), 1e-300)


def _segment_switches(r, it, nseg):
    """the switches of switch.F:228-242 for step `it` (1-based) of ocean segments of `nseg` steps, the counters the overlay
    predicts the next step's kind from (switch.F:217-223) and a clock that stands still (tmngr.F:330-367)"""
    r.ref.set("osegs", 1 if (it - 1) % nseg == 0 else 0)
    r.ref.set("osege", 1 if it % nseg == 0 else 0)
    r.ref.set("ntspos", nseg)
    r.ref.set("itt", it)
    r.ref.set("prelyr", float(r.v["relyr"][0]))


@pytest.mark.parametrize("exact", [True, False])
@pytest.mark.parametrize("cfg,dims,level", [("p2", (14, 14, 6), "1"), ("c30", (14, 14, 6), "1"), ("s37", (14, 14, 6), "1"),
                                            ("p2", (14, 14, 6), "3"), ("c30", (14, 14, 6), "3")])
def test_resident_overlay_over_several_steps(cfg, dims, level, exact, monkeypatch):
    """UVIC_RESIDENT=1: t stays on the device and rotates there (what loadmw/putmw and the
    ramdrive do on the host); per step only T and S come back, the call returns as soon as they have, the sources and
    the isopycnal tensor of the next step are started ahead inside an ocean segment, and the surface sums of set_sbc
    stay on the device until the segment's last step.  Six steps of the reference's own call sequence in two segments
    of three -- leapfrog, a forward (mixing) step the overlay did not foresee (its look-ahead must be dropped), and one
    step on which the overlay hands the work to the reference routine (every tracer goes down, t(tau+1) comes up) --
    against the unmodified reference.  level 3 (UVIC_RESIDENT=3, mixing_gpu.F): with the reference's vmixc (tidal mixing) in
    the loop, and its isopyc and vmixc left out by the overlay's side on the steps the device takes -- the forward step
    forms the tensor and diff_cbt inside the step, the handed-over step runs the host's two routines again."""
    _arith(monkeypatch, exact)
    monkeypatch.setenv("UVIC_RESIDENT", level)
    if not (refmodel.available(cfg, *dims) and refmodel.available(cfg, *dims, shim=True)):
        pytest.skip("oracle/_ref reference/shim libraries did not travel with the tree")
    import refdriver
    oc = synthetic.make_ocean(cfg, *dims)
    ref = refdriver.RefOcean(oc)
    shim = refdriver.RefOcean(oc, shim=True)
    if not hasattr(shim.ref.lib, "tracer_gpu_flush_"):
        pytest.skip("oracle/_ref shim predates the resident mode")
    if level == "3":
        if not hasattr(shim.ref.lib, "uvic_mix_on_host_"):
            pytest.skip("oracle/_ref shim predates mixing_gpu.F")
        tid = synthetic.make_tidal(oc.grid, oc.topo, oc.params.kappa_h)
        ref.set_tidal(tid); shim.set_tidal(tid)
    jmt = dims[1]
    # surface boundary conditions: T, S and (with MOBI) three more tracers feed sbc slots (csbc.h: trsbcindex)
    slots = {0: 3, 1: 4}
    if oc.cfg.nt > 2:
        slots.update({oc.cfg.tracers.index(n): 10 + q for q, n in enumerate(("dic", "o2", "alk"))})
    for r in (ref, shim):
        for n, k in slots.items():
            r.v["trsbcindex"][n] = k
    for it in range(1, 7):
        forward, on_host = it == 3, it == 5
        for r in (ref, shim):
            r.set_step_kind(forward)
            r.ref.set("euler2", 1 if on_host else 0)       # the overlay's cue to call the reference routine
            _segment_switches(r, it, 3)
        want = ref.step().copy()
        got = shim.step().copy()
        # what the host reads between two steps: T and S whole ...
        assert _same(got[:, :, 1:jmt - 1, :2], want[:, :, 1:jmt - 1, :2], exact), it
        if it % 3 == 0:     # ... and at a segment's last step the surface averages for the atmosphere
            for n, k in slots.items():
                # (T, S: bit for bit in the exact arithmetic; dic, o2, alk carry MOBI sources, whose device exp/log differ from libm)
                assert _same(shim.v["sbc"][1:-1, 1:jmt - 1, k - 1], ref.v["sbc"][1:-1, 1:jmt - 1, k - 1], exact and n < 2), (it, n)
        ref.rotate(); shim.rotate()
    if oc.cfg.nt > 2:   # resident for real: below the surface the host copy of the other tracers is stale until the flush
        assert not np.array_equal(shim.v["t"][:, 1:, 1:jmt - 1, 2:, 1], ref.v["t"][:, 1:, 1:jmt - 1, 2:, 1])
    shim.flush()
    for slot in (0, 1):                                      # t(tau-1), t(tau) of the coming step, every tracer
        a, b = shim.v["t"][..., slot], ref.v["t"][..., slot]
        assert _same(a[:, :, 1:jmt - 1, :2], b[:, :, 1:jmt - 1, :2], exact)
        for n, name in enumerate(oc.cfg.tracers):
            x, y = a[:, :, 1:jmt - 1, n], b[:, :, 1:jmt - 1, n]
            assert np.abs(x - y).max() <= PROD_TOL * np.abs(y).max(), (slot, name, np.abs(x - y).max())


@pytest.mark.parametrize("level", ["1", "3"])
@pytest.mark.parametrize("cfg,dims", [("c30", (14, 14, 6)), ("s37", (14, 14, 6)), ("c30", (102, 102, 19))])
def test_resident_overlay_takes_new_forcing_at_every_segment(cfg, dims, level, monkeypatch):
    """Between two ocean segments the atmosphere and the ice model change what MOBI reads of them -- the short-wave
    radiation, ice cover, ice and snow thickness (light under ice, tracer.F:355-420) and the atmospheric CO2 -- and the
    first step of the new segment computes its sources in line from the new fields (the overlay fetches them from the
    caller's arrays: k_pull4).  Three segments of three steps, every one with its own forcing, against the unmodified
    reference with the same changes; production arithmetic."""
    monkeypatch.delenv("UVIC_EXACT", raising=False)
    monkeypatch.setenv("UVIC_RESIDENT", level)
    if not (refmodel.available(cfg, *dims) and refmodel.available(cfg, *dims, shim=True)):
        pytest.skip("oracle/_ref reference/shim libraries did not travel with the tree")
    import refdriver
    oc = synthetic.make_ocean(cfg, *dims)
    ref = refdriver.RefOcean(oc)
    shim = refdriver.RefOcean(oc, shim=True)
    if not hasattr(shim.ref.lib, "tracer_gpu_flush_"):
        pytest.skip("oracle/_ref shim predates the resident mode")
    if level == "3":
        if not hasattr(shim.ref.lib, "uvic_mix_on_host_"):
            pytest.skip("oracle/_ref shim predates mixing_gpu.F")
        tid = synthetic.make_tidal(oc.grid, oc.topo, oc.params.kappa_h)
        ref.set_tidal(tid); shim.set_tidal(tid)
    imt, jmt = dims[0], dims[1]
    f = oc.forcing
    ii, jj = np.meshgrid(np.arange(imt), np.arange(jmt), indexing="ij")
    for it in range(1, 10):
        seg = (it - 1) // 3
        for r in (ref, shim):
            r.set_step_kind(False)
            _segment_switches(r, it, 3)
            if (it - 1) % 3 == 0:      # what the coupler does between segments (the same for both)
                r.v["dnswr"][...] = f.dnswr * (1.0 + 0.25 * seg) * (1.0 + 0.1 * np.sin(0.3 * ii + seg))
                r.v["aice"][:, :, 1] = np.clip(f.aice + 0.3 * seg * (np.cos(0.2 * jj) > 0.5), 0.0, 1.0)
                r.v["hice"][:, :, 1] = f.hice + 20.0 * seg * (r.v["aice"][:, :, 1] > 0)
                r.v["hsno"][:, :, 1] = f.hsno + 5.0 * seg * (r.v["aice"][:, :, 1] > 0)
                r.ref.set("co2ccn", f.co2ccn + 40.0 * seg)
        want = ref.step().copy()
        got = shim.step().copy()
        assert np.array_equal(got[:, :, 1:jmt - 1, :2], want[:, :, 1:jmt - 1, :2]), it
        ref.rotate(); shim.rotate()
    shim.flush()
    a, b = np.array(shim.v["t"][..., 1]), np.array(ref.v["t"][..., 1])     # (copies: one library = one set of COMMON blocks)
    for n, name in enumerate(oc.cfg.tracers):
        x, y = a[:, :, 1:jmt - 1, n], b[:, :, 1:jmt - 1, n]
        assert np.abs(x - y).max() <= PROD_TOL * np.abs(y).max(), (name, np.abs(x - y).max())
    # the forcing mattered: the same run with the first segment's forcing throughout ends elsewhere
    ref2 = refdriver.RefOcean(oc)
    if level == "3":
        ref2.set_tidal(tid)
    for it in range(1, 10):
        ref2.set_step_kind(False)
        _segment_switches(ref2, it, 3)
        ref2.step()
        ref2.rotate()
    assert not np.allclose(ref2.v["t"][:, 0, 1:jmt - 1, 2:, 1], b[:, 0, 1:jmt - 1, 2:], rtol=1e-6, atol=0.0)


@pytest.mark.parametrize("seed", range(1, 1 + int(__import__("os").environ.get("UVIC_TEST_SCHEDULES", "6"))))
@pytest.mark.parametrize("cfg", ["c30", "s37"])
@pytest.mark.parametrize("level", ["0", "1", "3"])
def test_resident_overlay_random_schedules(level, cfg, seed, monkeypatch):
    """`tracer` alone, sixteen steps whose kinds are drawn at random (fixed seeds): forward steps seen coming or not (nmix),
    steps handed to the reference routine, time-average steps, segments of random length each with its own atmosphere and
    ice fields.  T and S of every step bit for bit, every tracer after the last step to the production tolerance."""
    monkeypatch.delenv("UVIC_EXACT", raising=False)
    if level == "0":      # the default: nothing resident
        monkeypatch.delenv("UVIC_RESIDENT", raising=False)
    else:
        monkeypatch.setenv("UVIC_RESIDENT", level)
    dims = (14, 14, 6)
    if not (refmodel.available(cfg, *dims) and refmodel.available(cfg, *dims, shim=True)):
        pytest.skip("oracle/_ref reference/shim libraries did not travel with the tree")
    import refdriver
    oc = synthetic.make_ocean(cfg, *dims)
    ref = refdriver.RefOcean(oc)
    shim = refdriver.RefOcean(oc, shim=True)
    if not hasattr(shim.ref.lib, "uvic_mix_on_host_"):
        pytest.skip("oracle/_ref shim predates mixing_gpu.F")
    tid = synthetic.make_tidal(oc.grid, oc.topo, oc.params.kappa_h)
    ref.set_tidal(tid); shim.set_tidal(tid)
    rng = np.random.default_rng(77 * seed + 5)
    nsteps, seg, nmix = 16, int(rng.integers(2, 6)), int(rng.integers(0, 5))
    forward = rng.random(nsteps + 1) < 0.2
    on_host = rng.random(nsteps + 1) < 0.15
    tavg = rng.random(nsteps + 1) < 0.2
    imt, jmt = dims[0], dims[1]
    f = oc.forcing
    ii, jj = np.meshgrid(np.arange(imt), np.arange(jmt), indexing="ij")
    # the model's clock advances every step (tmngr.F:330-367); the month of the dust field and the solar declination follow it
    # (tracer.F:311-338), and the overlay computes the next step's sources ahead with the clock it expects then: start near a
    # month boundary, near the end of the year (relyr wraps: the guess is wrong once and must be caught), or anywhere
    dyr = oc.params.dtts / (365.0 * 86400.0)
    start = [1.0 / 12.0 - 3.3 * dyr, 1.0 - 4.2 * dyr, 0.4][int(rng.integers(0, 3))]
    what = dict(seg=seg, nmix=nmix, forward=np.flatnonzero(forward).tolist(), on_host=np.flatnonzero(on_host).tolist(),
                tavg=np.flatnonzero(tavg).tolist(), clock=start)
    for it in range(1, nsteps + 1):
        k = (it - 1) // seg
        for r in (ref, shim):
            r.set_step_kind(bool(forward[it]))
            r.ref.set("euler2", 1 if on_host[it] else 0)
            r.ref.set("timavgperts", 1 if tavg[it] else 0)
            _segment_switches(r, it, seg)
            r.ref.set("nmix", nmix)
            r.ref.set("prelyr", (start + (it - 2) * dyr) % 1.0 if it > 1 else (start - dyr) % 1.0)
            r.v["relyr"][...] = (start + (it - 1) * dyr) % 1.0
            if (it - 1) % seg == 0:
                r.v["dnswr"][...] = f.dnswr * (1.0 + 0.1 * (k % 3)) * (1.0 + 0.1 * np.sin(0.3 * ii + k))
                r.v["aice"][:, :, 1] = np.clip(f.aice + 0.2 * (k % 3) * (np.cos(0.2 * jj) > 0.5), 0.0, 1.0)
                r.v["hice"][:, :, 1] = f.hice + 10.0 * (k % 3) * (r.v["aice"][:, :, 1] > 0)
                r.ref.set("co2ccn", f.co2ccn + 10.0 * k)
        want = ref.step().copy()
        got = shim.step().copy()
        assert np.array_equal(got[:, :, 1:jmt - 1, :2], want[:, :, 1:jmt - 1, :2]), (it, what)
        ref.rotate(); shim.rotate()
    if level != "0":
        shim.flush()
    a, b = np.array(shim.v["t"][..., 1]), np.array(ref.v["t"][..., 1])
    for n, name in enumerate(oc.cfg.tracers):
        x, y = a[:, :, 1:jmt - 1, n], b[:, :, 1:jmt - 1, n]
        assert np.abs(x - y).max() <= PROD_TOL * np.abs(y).max(), (name, np.abs(x - y).max(), what)


def test_resident_overlay_back_to_back_on_the_full_grid(monkeypatch):
    """The resident overlay on 102x102x19 with option set C, sixteen steps in segments of four with NOTHING between the calls
    but the reference's own step -- long enough kernels and a busy enough device for the overlay's asynchronous schedule to
    show if it were wrong: the step's inputs arrive by copies beside the main stream into device copies taken in turn, the
    T,S chain of a step starts while the other tracers of the step before are still in their pass B, MOBI sources of a
    segment's first step run beside pass A.  Production arithmetic: T and S of EVERY step bit-identical to the unmodified
    reference, every tracer after the flush to the production tolerance."""
    monkeypatch.delenv("UVIC_EXACT", raising=False)
    monkeypatch.setenv("UVIC_RESIDENT", "1")
    cfg, dims = "c30", (102, 102, 19)
    if not (refmodel.available(cfg, *dims) and refmodel.available(cfg, *dims, shim=True)):
        pytest.skip("oracle/_ref reference/shim libraries did not travel with the tree")
    import refdriver
    oc = synthetic.make_ocean(cfg, *dims)
    ref = refdriver.RefOcean(oc)
    shim = refdriver.RefOcean(oc, shim=True)
    jmt = dims[1]
    rng = np.random.default_rng(11)
    for r in (ref, shim):
        r.set_step_kind(False)
        r.ref.set("nmix", 0)
    for it in range(1, 17):
        stf = oc.stf * (1.0 + 0.02 * it) + 1e-9 * rng.standard_normal(oc.stf.shape) * (oc.topo.kmt > 0)[..., None]
        # the inputs change from step to step; adv_vbt by continuity from adv_vet, adv_vnt, as adv_vel.F makes it
        vet, vnt, vbt = synthetic.advective_velocities(oc.grid, oc.u * (1.0 + 0.03 * np.sin(0.9 * it)))
        for r in (ref, shim):
            _segment_switches(r, it, 4)
            r.v["stf"][...] = stf
            r.v["adv_vet"][...] = vet[:, :, 1:]
            r.v["adv_vnt"][...] = vnt
            r.v["adv_vbt"][...] = vbt[:, :, 1:]
        want = ref.step().copy()
        got = shim.step().copy()
        assert np.array_equal(got[:, :, 1:jmt - 1, :2], want[:, :, 1:jmt - 1, :2]), it
        ref.rotate(); shim.rotate()
    shim.flush()
    a, b = shim.v["t"][..., 1], ref.v["t"][..., 1]
    for n, name in enumerate(oc.cfg.tracers):
        x, y = a[:, :, 1:jmt - 1, n], b[:, :, 1:jmt - 1, n]
        assert np.abs(x - y).max() <= PROD_TOL * np.abs(y).max(), (name, np.abs(x - y).max())


@pytest.mark.parametrize("exact", [True, False])
def test_resident_overlay_through_an_euler_backward_step(exact, monkeypatch):
    """An Euler backward mixing step (eb): both passes run on the host (euler1 with eots false, then euler2), after which
    the reference shuffles its time levels (source/mom/odam.F:251, mom.F:434-446) instead of rotating them.  The overlay
    must not rotate the device's levels behind a step it did not take: it gives the state back and takes t(tau-1), t(tau)
    from the host at the next leapfrog step."""
    _arith(monkeypatch, exact)
    monkeypatch.setenv("UVIC_RESIDENT", "1")
    cfg, dims = "c30", (14, 14, 6)
    if not (refmodel.available(cfg, *dims) and refmodel.available(cfg, *dims, shim=True)):
        pytest.skip("oracle/_ref reference/shim libraries did not travel with the tree")
    import refdriver
    oc = synthetic.make_ocean(cfg, *dims)
    ref = refdriver.RefOcean(oc)
    shim = refdriver.RefOcean(oc, shim=True)
    jmt = dims[1]

    def leap(it):
        for r in (ref, shim):
            r.set_step_kind(False)
            r.ref.set("euler1", 0); r.ref.set("euler2", 0); r.ref.set("eots", 1)
            _segment_switches(r, it, 100)
        want, got = ref.step().copy(), shim.step().copy()
        assert _same(got[:, :, 1:jmt - 1, :2], want[:, :, 1:jmt - 1, :2], exact), it
        ref.rotate(); shim.rotate()

    leap(1); leap(2)
    # The Euler backward step as mom.F drives it.  First pass: forward from tau, eots false.  (set_step_kind copies the
    # host's tau into the tau-1 slot; in the resident overlay that host copy is stale and the overlay brings the device's in.)
    for r in (ref, shim):
        r.set_step_kind(True)
        r.ref.set("forward", 0); r.ref.set("euler1", 1); r.ref.set("euler2", 0); r.ref.set("eots", 0)
        _segment_switches(r, 3, 100)
    w1, g1 = ref.step().copy(), shim.step().copy()
    assert _same(g1[:, :, 1:jmt - 1, :2], w1[:, :, 1:jmt - 1, :2], exact)      # (the other tracers carry the device MOBI's rounding)
    assert _same(g1[:, :, 1:jmt - 1], w1[:, :, 1:jmt - 1], False)
    # second pass: the first guess is tau, tau-1 still the state before the step; eots true
    for r in (ref, shim):
        t = r.v["t"]
        t[..., 1] = t[..., 2]
        r.ref.set("euler1", 0); r.ref.set("euler2", 1); r.ref.set("eots", 1)
    w2, g2 = ref.step().copy(), shim.step().copy()
    assert _same(g2[:, :, 1:jmt - 1, :2], w2[:, :, 1:jmt - 1, :2], exact)
    assert _same(g2[:, :, 1:jmt - 1], w2[:, :, 1:jmt - 1], False)
    # the host re-points its levels (not a rotation): tau-1 stays the state before the mixing step, tau is the result
    for r in (ref, shim):
        t = r.v["t"]
        t[..., 1] = t[..., 2]
    # from here the device must have the host's tau-1 and tau, not a rotation of what it held
    leap(4); leap(5)
    shim.flush()
    for slot in (0, 1):
        a, b = shim.v["t"][..., slot], ref.v["t"][..., slot]
        for n, name in enumerate(oc.cfg.tracers):
            x, y = a[:, :, 1:jmt - 1, n], b[:, :, 1:jmt - 1, n]
            assert np.abs(x - y).max() <= (0.0 if exact and n < 2 else PROD_TOL) * np.abs(y).max(), (slot, name)


@pytest.mark.parametrize("resident", [False, True])
@pytest.mark.parametrize("cfg,dims", [("m2", (14, 14, 6)), ("m2", (102, 102, 19)), ("m2i", (14, 14, 6))])
def test_overlay_clinic_matches_reference_clinic(cfg, dims, resident, monkeypatch):
    """The momentum row's boundary end to end: `clinic(joff,js,je,is,ie)` of the package's
    overlay (uvic2.9_amd/fortran/clinic_gpu.F) against the reference's own routine, through the reference's COMMON
    blocks: u(tau+1), zu and the four sbc planes of isbcu/asbcu, bit for bit, with the polar filter filuv on.  The
    overlay shares the device instance of the `tracer` overlay, which runs first as in mom.F:389-395.  (m2i: a build
    without O_anisotropic_viscosity and O_ice_evp -- the overlay spreads the per-row coefficients itself.)"""
    monkeypatch.delenv("UVIC_EXACT", raising=False)
    if not (refmodel.available(cfg, *dims) and refmodel.available(cfg, *dims, shim=True)):
        pytest.skip("oracle/_ref reference/shim libraries did not travel with the tree")
    import refdriver
    oc = synthetic.make_ocean(cfg, *dims)
    mom = synthetic.make_momentum(oc.grid, oc.topo, oc.u, anisotropic=(cfg == "m2"))
    flt = synthetic.make_filter_u(oc.grid, dims[2])
    rng = np.random.default_rng(3)
    planes = rng.standard_normal((dims[0], dims[1], 4))

    def prepare(R):
        R.set_momentum(mom)
        R.set_filter(synthetic.make_filter(oc.grid, dims[2]))     # index.h holds the rows of both filters (setcom.F:75-86)
        R.set_filter_u(flt)
        S, v = R.ref.set, R.v
        S("igu", 11); S("igv", 12); S("isu", 13); S("isv", 14); S("ntspos", 3)
        v["sbc"][:, :, 10:14] = planes
        R.state()
        R.adv_vel_u()
        R.setvbc()

    ref = refdriver.RefOcean(oc)
    prepare(ref)
    shim = refdriver.RefOcean(oc, shim=True)
    prepare(shim)
    # before `tracer` has made the device instance (a run that opens with an Euler backward step or a diagnostic step, on
    # which the tracer overlay falls back) the clinic overlay hands the step to the reference routine as well
    want_u, want_zu, _ = ref.clinic()
    got_u, got_zu, _ = shim.clinic()
    assert np.array_equal(got_u, want_u) and np.array_equal(got_zu, want_zu)
    shim.step()                      # isopyc, "+K33", tracer (overlay): creates the device instance, sends adv_v?t
    for itt, (osegs, osege) in enumerate(((1, 0), (0, 0), (0, 1))):
        for R in (ref, shim):
            R.ref.set("osegs", osegs); R.ref.set("osege", osege)
            R.ref.set("itt", itt)    # from the second pass on the overlay sends the advective velocities itself
        want_u, want_zu, _ = ref.clinic()
        got_u, got_zu, _ = shim.clinic()
        assert np.array_equal(got_u[:, :, 1:-1], want_u[:, :, 1:-1])
        assert np.array_equal(got_zu, want_zu)
        assert np.array_equal(shim.v["sbc"][:, :, 10:14], ref.v["sbc"][:, :, 10:14])
    assert np.abs(want_u).max() > 0.1
    # a diagnostic time step goes through the reference routine, kept as clinic_cpu
    for R in (ref, shim):
        R.ref.set("tsiperts", 1)
    want_u, want_zu, _ = ref.clinic()
    got_u, got_zu, _ = shim.clinic()
    assert np.array_equal(got_u, want_u) and np.array_equal(got_zu, want_zu)


@pytest.mark.parametrize("exact", [True, False])
@pytest.mark.parametrize("resident", [False, True])
def test_time_step_monitor_steps_stay_on_the_device(resident, exact, monkeypatch):
    """With the shipped run/control.in (tsiint = tsiper = 10 days) tsiperts (source/common/switch.F:458-459) is true on
    EVERY ocean step and run/mk.in defines O_time_step_monitor: `tracer` then also forms tbar, travar, dtabs (diagt1,
    u09/mom/tracer.F:1516-1537; dtabs from t(tau+1) BEFORE convection) and dc14bar (:1329-1353), `clinic` ektot
    (clinic.F:616-630).  The overlays keep such steps on the device and hand the same COMMON arrays to diago: three
    steps (leapfrog, leapfrog, forward) of option set C built with the time-step monitor against the unmodified
    reference -- the integrals bit for bit in the exact arithmetic (dc14bar, a single running sum over the whole grid in
    the reference, to rounding), to the production tolerance otherwise."""
    _arith(monkeypatch, exact)
    if resident:
        monkeypatch.setenv("UVIC_RESIDENT", "1")
    else:
        monkeypatch.delenv("UVIC_RESIDENT", raising=False)
    cfg, dims = "t30", (14, 14, 6)
    if not (refmodel.available(cfg, *dims) and refmodel.available(cfg, *dims, shim=True)):
        pytest.skip("oracle/_ref reference/shim libraries did not travel with the tree")
    import refdriver
    oc = synthetic.make_ocean(cfg, *dims)
    mom = synthetic.make_momentum(oc.grid, oc.topo, oc.u, anisotropic=True)
    ref = refdriver.RefOcean(oc)
    shim = refdriver.RefOcean(oc, shim=True)
    jmt = dims[1]
    for R in (ref, shim):
        R.set_momentum(mom)
        R.ref.set("tsiperts", 1)
        R.state(); R.adv_vel_u(); R.setvbc()
    names = ("tbar", "travar", "dtabs")
    for it in range(1, 4):
        forward = it == 3
        for R in (ref, shim):
            R.set_step_kind(forward)
            _segment_switches(R, it, 4)
            for n in names + ("ektot",):        # diagi zeroes them at the start of every step (source/mom/diagi.F:193-205)
                R.v[n][...] = 0.0
            R.ref.set("dc14bar", 0.0)
        want = ref.step().copy()
        got = shim.step().copy()
        assert _same(got[:, :, 1:jmt - 1, :2], want[:, :, 1:jmt - 1, :2], exact), it
        if not exact and it == 1:   # the device did this step: the other tracers carry the column kernels' rounding
            assert not np.array_equal(got[:, :, 1:jmt - 1, 2:], want[:, :, 1:jmt - 1, 2:])
        for n in names:
            a, b = shim.v[n], ref.v[n]
            assert np.abs(b).max() > 0, n
            if exact:
                # (T, S bit for bit; the MOBI tracers through the device's exp/log: to rounding)
                assert np.array_equal(a[:, :2], b[:, :2]), (it, n)
            scale = np.abs(b).max(axis=(0, 2), keepdims=True)
            assert (np.abs(a - b) <= PROD_TOL * np.maximum(scale, 1e-300)).all(), (it, n)
        d_got, d_want = float(shim.v["dc14bar"][0]), float(ref.v["dc14bar"][0])
        assert d_want != 0.0 and abs(d_got - d_want) <= PROD_TOL * abs(d_want), (it, d_got, d_want)
        # clinic of the same step: ektot from u(tau)
        want_u, _, _ = ref.clinic()
        got_u, _, _ = shim.clinic()
        assert np.array_equal(got_u[:, :, 1:-1], want_u[:, :, 1:-1])
        assert np.abs(ref.v["ektot"]).max() > 0 and np.array_equal(shim.v["ektot"], ref.v["ektot"]), it
        ref.rotate(); shim.rotate()


@pytest.mark.parametrize("exact", [True, False])
@pytest.mark.parametrize("resident", [False, True])
def test_time_average_steps_stay_on_the_device(resident, exact, monkeypatch):
    """With the shipped run/control.in (timavgint = 3650 d, timavgper = 365 d) timavgperts is true on one year in ten, and
    run/mk.in defines O_time_averages and O_save_convection: on such a step `tracer` also adds the convection diagnostics of
    convct2 (totalk, vdepth, pe; convect.F:183-301, tracer.F:1211-1222) and the delta-14C field (tracer.F:1329-1364) to its
    time averages, and `diag -> avgvar` afterwards reads t(tau) of EVERY tracer from the memory window.  The overlay keeps
    such steps on the device (a step through the reference routine costs a thousand device steps): five steps of option set C,
    three of them time-average steps and all of them time-step-monitor steps, against the unmodified reference -- the
    accumulated ta_totalk, ta_vdepth, ta_pe, nta_conv bit for bit (they depend on T and S alone), ta_dc14 and the host's
    t(tau) on those steps to the production tolerance."""
    _arith(monkeypatch, exact)
    if resident:
        monkeypatch.setenv("UVIC_RESIDENT", "1")
    else:
        monkeypatch.delenv("UVIC_RESIDENT", raising=False)
    cfg, dims = "t30", (14, 14, 6)
    if not (refmodel.available(cfg, *dims) and refmodel.available(cfg, *dims, shim=True)):
        pytest.skip("oracle/_ref reference/shim libraries did not travel with the tree")
    import refdriver
    oc = synthetic.make_ocean(cfg, *dims)
    ref = refdriver.RefOcean(oc)
    shim = refdriver.RefOcean(oc, shim=True)
    if not hasattr(shim.ref.lib, "tracer_gpu_tavg_"):
        pytest.skip("oracle/_ref shim predates the time-average steps on the device")
    jmt = dims[1]
    names = ("ta_totalk", "ta_vdepth", "ta_pe", "ta_dc14")
    for R in (ref, shim):
        R.set_step_kind(False)
        R.ref.set("nmix", 0); R.ref.set("tsiperts", 1); R.ref.set("nta_conv", 0)
        for n in names:
            R.v[n][...] = 0.0
        # a few columns with dense water on top, so that the convective walk has something to do on every step
        t = R.v["t"]
        for slot in (0, 1):
            t[3:9, 0, 3:9, 0, slot] -= 6.0
            t[3:9, 0, 3:9, 0, slot] *= oc.topo.tmask[3:9, 0, 3:9]
    for it in range(1, 6):
        tavg = it in (2, 3, 4)
        for R in (ref, shim):
            _segment_switches(R, it, 4)
            R.ref.set("timavgperts", 1 if tavg else 0)
            for n in ("tbar", "travar", "dtabs"):
                R.v[n][...] = 0.0
            R.ref.set("dc14bar", 0.0)
        want = ref.step().copy()
        got = shim.step().copy()
        assert _same(got[:, :, 1:jmt - 1, :2], want[:, :, 1:jmt - 1, :2], exact), it
        if tavg:    # what avgvar reads next: t(tau) of every tracer, on the host
            a, b = shim.v["t"][:, :, 1:jmt - 1, :, 1], ref.v["t"][:, :, 1:jmt - 1, :, 1]
            for n, name in enumerate(oc.cfg.tracers):
                assert np.abs(a[..., n] - b[..., n]).max() <= PROD_TOL * max(np.abs(b[..., n]).max(), 1e-300), (it, name)
        ref.rotate(); shim.rotate()
    assert int(shim.ref.get("nta_conv")) == int(ref.ref.get("nta_conv")) == 3
    assert ref.v["ta_totalk"].max() >= 2.0 and np.abs(ref.v["ta_pe"]).max() > 0.0 and ref.v["ta_vdepth"].max() > 0.0
    for n in ("ta_totalk", "ta_vdepth", "ta_pe"):
        assert np.array_equal(shim.v[n], ref.v[n]), n
    a, b = shim.v["ta_dc14"], ref.v["ta_dc14"]
    assert np.abs(b).max() > 0 and np.abs(a - b).max() <= PROD_TOL * np.abs(b).max()
