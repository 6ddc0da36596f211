#!/usr/bin/env python3
"""bench.py -- tracer-cell updates/s of the UVic 2.9 ocean tracer step on MI355X.

    python bench.py --gpus N --steps K --warmup W [--cfg c30|p2|perfNN] [--grid 102x102x19]

One "step" is one pass of the hot path over the whole synthetic ocean: isopyc
(mixing tensor + GM velocities), the tracer step (MOBI sources when the option
set has them, FCT advection, isopycnal diffusion, explicit update, implicit
vertical solve, convection) and the time-level rotation, with every input
already resident in HBM.  Metric (BASELINE.json): imt*jmt*km*nt cell updates
per completed step / wall time, whole job.  N > 1 (launched by
torch.distributed.run, one rank per GPU, RCCL): tracer-index shards with an all-gather of
t(tau+1) per step on the default grid (BASELINE configs 3-4), latitude slabs with a 2-row halo
exchange on the refined 202x202x32 grid (config 5) (--decomp overrides).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
for p in (ROOT, ROOT / "oracle"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s achievable)


def b_alg(nt, nsrc):
    """Algorithmic bytes per cell update of the whole step, SURVEY.md §8(d)."""
    return 24.0 + 8.0 * nsrc / nt + 280.0 / nt


def kernel_alg_bytes(name, nt, nsrc):
    """Algorithmic bytes per cell update of ONE kernel launch:
    what the kernel must read/write once if every stencil neighbour were free."""
    if name == "fct_rows":       # reads t(tau-1), t(tau); shared: 3 total velocities, tmask; writes nothing algorithmic
        return 16.0 + 32.0 / nt
    if name == "update_rows":    # reads t(tau-1), t(tau), source; writes t(tau+1); shared isopyc fields + metrics
        return 24.0 + 8.0 * nsrc / nt + 248.0 / nt
    if name == "colfct":         # reads t(tau-1), t(tau); shared: 3 total velocities, 19 folded coefficients, kmt
        return 16.0 + 184.0 / nt
    if name == "colupd":         # reads t(tau-1), t(tau), source; writes t(tau+1); shared: tot_n, diff_cbt
        return 24.0 + 8.0 * nsrc / nt + 16.0 / nt
    return None


def pmc_traffic(kernel, workload_ok, split=False):
    """HBM bytes per launch of `kernel` from the rocprofv3 --pmc passes of tools/profile_round.sh
    (FETCH_SIZE + WRITE_SIZE in separate passes, corrected by the 8-B-per-lane calibration of
    tools/calib_traffic.hip as MI355X_MICROARCH.md prescribes).  The counters cannot be read from
    inside this process: the figure comes from the committed summary of the SAME command
    (profiles/traffic.json, or $UVIC_TRAFFIC_JSON) and is null for any other workload."""
    path = os.environ.get("UVIC_TRAFFIC_JSON") or str(ROOT / "profiles" / "traffic.json")
    if not workload_ok or not os.path.exists(path):
        return None, None, None
    try:
        kernels = json.load(open(path))["kernels"]
        # the mark "colfct" / "colupd" is the pass, whichever form of the kernel the library launched for it
        key = "k_" + kernel
        if split:   # the summary lists a kernel per grid: plain name = all tracers in one launch (mixing steps, isolated
            # profile), "name#<grid>" the others; the main-stream launch of the other nt-2 tracers is the largest of those
            alt = [k for k in kernels if k.startswith(key + "#")]
            if alt:
                key = max(alt, key=lambda k: int(k.split("#")[1]))
        ent = kernels.get(key)
        return (ent or {}).get("total"), os.path.relpath(path, ROOT), (ent or {}).get("valu_wave_insts")
    except (ValueError, KeyError, OSError):
        return None, None, None


def hbm_probe(torch, mib=1024, nrep=20):
    """The HBM denominator measured on this box: a device-to-device copy and a triad a = b + s*c over
    `mib` MiB arrays (far beyond the 256 MiB Infinity Cache), bytes moved / time, HIP events on the current stream."""
    n = mib * (1 << 20) // 8
    a = torch.empty(n, dtype=torch.float64, device="cuda")
    b = torch.ones(n, dtype=torch.float64, device="cuda")
    c = torch.ones(n, dtype=torch.float64, device="cuda")
    out = {}
    for name, fn, nbytes in (("copy", lambda: a.copy_(b), 2 * n * 8), ("triad", lambda: torch.add(b, c, alpha=0.5, out=a), 3 * n * 8)):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(nrep):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out[name + "_GBs"] = nbytes * nrep / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del a, b, c
    torch.cuda.empty_cache()
    out["arrays_MiB"] = mib
    return out


def cpu_baseline(ocean, to, so, c, src, budget_s=20.0):
    """Time the CPU path on this box's host cores, 1 core, bounded sample."""
    import oracle_c
    import refmodel
    g, cfg = ocean.grid, ocean.cfg
    units = g.imt * g.jmt * g.km * cfg.nt
    if cfg.name in ("p2", "c30") and refmodel.available(cfg.name, g.imt, g.jmt, g.km):
        try:
            import refdriver
            ro = refdriver.RefOcean(ocean)
            n, t0 = 0, time.perf_counter()
            while True:
                ro.step()
                n += 1
                el = time.perf_counter() - t0
                if el > budget_s or n >= 5:
                    break
            return {"value": units * n / el, "unit": "cell-updates/s", "cores": 1, "kind": "reference",
                    "sample": f"{n} full isopyc+tracer steps of the compiled reference Fortran (oracle/_ref, flang -O2), "
                              f"{cfg.name} {g.imt}x{g.jmt}x{g.km}"}
        except Exception as e:  # fall through to the C port
            print(f"[bench] reference baseline unavailable: {e}", file=sys.stderr)
    orc = oracle_c.Oracle(ocean, to=to, so=so, c=c, src=src)
    prm = None
    if cfg.ntnpzd:
        import mobi_c
        from uvic29_amd import mobi as pm
        prm = pm.load_table(cfg.name, g.km)
    n, t0 = 0, time.perf_counter()
    while True:
        if prm is not None:
            orc.set_src(mobi_c.mobi_sources(ocean, prm, ocean.t_taum1, 2.0 * ocean.params.dtts))
        orc.isopyc(); orc.add_k33(); orc.transport()
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 5:
            break
    return {"value": units * n / el, "unit": "cell-updates/s", "cores": 1, "kind": "port",
            "sample": f"{n} full steps (isopyc, {'MOBI sources, ' if prm is not None else ''}transport, convection) of the C "
                      f"oracle (gcc -O2 -ffp-contract=off), {cfg.name} {g.imt}x{g.jmt}x{g.km}"}


def cpu_baseline_ncore(ocean, to, so, c, src, budget_s=8.0):
    """A courtesy figure beside `cpu_baseline`: the C oracle with its independent pieces shared out over this box's cores
    (OpenMP: the tracers of the transport, the rows of convct2 and of the MOBI sources; isopyc stays on one).  The same
    arithmetic as on one core; what a maintainer could get from the host without touching the algorithm."""
    import oracle_c
    g, cfg = ocean.grid, ocean.cfg
    cores = max(1, min(16, len(os.sched_getaffinity(0))))
    lib = oracle_c.lib()
    lib.orc_set_threads(cores)
    try:
        orc = oracle_c.Oracle(ocean, to=to, so=so, c=c, src=src)
        prm = None
        if cfg.ntnpzd:
            import mobi_c
            from uvic29_amd import mobi as pm
            prm = pm.load_table(cfg.name, g.km)
        n, t0 = 0, time.perf_counter()
        while True:
            if prm is not None:
                orc.set_src(mobi_c.mobi_sources(ocean, prm, ocean.t_taum1, 2.0 * ocean.params.dtts))
            orc.isopyc(); orc.add_k33(); orc.transport()
            n += 1
            el = time.perf_counter() - t0
            if el > budget_s or n >= 20:
                break
    finally:
        lib.orc_set_threads(1)
    return {"value": g.imt * g.jmt * g.km * cfg.nt * n / el, "unit": "cell-updates/s", "cores": cores, "kind": "port",
            "sample": f"{n} full steps of the C oracle (gcc -O2 -ffp-contract=off -fopenmp), tracers / rows shared out over "
                      f"{cores} threads, {cfg.name} {g.imt}x{g.jmt}x{g.km}"}


def clinic_cpu_baseline(imt, jmt, km, ncall=5):
    """The reference's own state + clinic (with filuv) on one host core (oracle/_ref build "m2"), ms per call."""
    import refmodel
    if not refmodel.available("m2", imt, jmt, km):
        return None
    import refdriver
    from uvic29_amd import synthetic
    oc = synthetic.make_ocean("m2", imt, jmt, km)
    ro = refdriver.RefOcean(oc)
    ro.set_momentum(synthetic.make_momentum(oc.grid, oc.topo, oc.u))
    ro.set_filter_u(synthetic.make_filter_u(oc.grid, km))
    ro.adv_vel_u(); ro.setvbc()
    ro.state(); ro.clinic()
    g = oc.grid
    t0 = time.perf_counter()
    for _ in range(ncall):
        ro.ref.call("clinic", 0, 2, g.jmt - 1, 2, g.imt - 1)
    return (time.perf_counter() - t0) / ncall * 1e3


def overlay_baseline(ocean, steps=40, segment=4):
    """The path the north star names: the reference's own call sequence with `tracer` replaced by the Fortran overlay
    (uvic2.9_amd/fortran/tracer_gpu.F -> ISO_C_BINDING -> libuvic_gpu.so), resident mode, driven through the compiled
    reference's COMMON blocks (oracle/_ref/libuvicshim_*; tools/overlay_time.py in a process of its own).  PCIe included:
    velocities, diff_cbt, stf, btf up and T,S down every step; ocean segments of `segment` steps (run/control.in:
    segtim 5 d / dtts 1.25 d), whose first step computes its MOBI sources in line.
      call_ms      median wall time of one `tracer` call when the host does its own work between calls (here: the host
                   isopyc of the tracer-only integration), i.e. what the Fortran driver waits for per step
      loop_ms      wall time per step of the calls alone, back to back (the device never idles)"""
    import refmodel
    g, cfg = ocean.grid, ocean.cfg
    if cfg.name != "c30" or (g.imt, g.jmt, g.km) != (102, 102, 19) or not refmodel.available(cfg.name, g.imt, g.jmt, g.km, shim=True):
        return None
    return _tool_json("overlay_time.py", [str(steps)], {"UVIC_RESIDENT": "1", "OVERLAY_SEGMENT": str(segment)})


def _tool_json(script, args, env):
    """Run one of the Fortran-boundary timing tools in a process of its own -- as the Fortran driver is: no PyTorch, the
    library the first user of the HIP runtime (it then asks for the hardware queues its streams need) -- and read the JSON
    line it prints."""
    import subprocess
    r = subprocess.run([sys.executable, str(ROOT / "tools" / script), *args, "--json"], env=dict(os.environ, **env),
                       capture_output=True, text=True, timeout=600, cwd=str(ROOT))
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        raise RuntimeError((r.stderr or r.stdout)[-400:])
    return json.loads(lines[-1])


def ocean_overlay_baseline(imt, jmt, km, steps=32):
    """Both Fortran overlays in the reference's own order (`tracer` then `clinic`, source/mom/mom.F:389-395) on option set C
    built as run/mk.in builds it (oracle/_ref shim "t30": + O_stream_function, O_anisotropic_viscosity, O_ice_evp,
    O_time_step_monitor) with the switches the shipped run/control.in gives: tsiperts on every step (the time-step
    integrals are formed on the device), ocean segments of four steps; tracers and velocities resident and `isopyc` and
    `vmixc` left to the device (UVIC_RESIDENT=3, mixing_gpu.F: their two calls return at once -- isopyc_vmixc_calls_ms --
    where the reference's own routines take host_isopyc_vmixc_ms, measured the same way with UVIC_RESIDENT=2).
    The other host routines of the loop run between the calls and are not timed.  Medians, ms per call (tools/ocean_overlay_time.py)."""
    import refmodel
    if (imt, jmt, km) != (102, 102, 19) or not refmodel.available("t30", imt, jmt, km, shim=True):
        return None
    out = _tool_json("ocean_overlay_time.py", [str(steps), "t30"], {"UVIC_RESIDENT": "3"})
    host = _tool_json("ocean_overlay_time.py", ["12", "t30"], {"UVIC_RESIDENT": "2"})
    out["host_isopyc_vmixc_ms"] = host["isopyc_vmixc_calls_ms"]
    out["host_adv_vel_ms"] = host["adv_vel_call_ms"]
    out["host_state_ms"] = host["state_call_ms"]
    out["switches"] = "tsiperts every step (run/control.in: tsiint = tsiper), segments of 4 steps, UVIC_RESIDENT=3"
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--cfg", default="c30")
    ap.add_argument("--grid", default="102x102x19")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-overlay", action="store_true", help="skip the Fortran-overlay measurement (oracle/_ref/libuvicshim_*)")
    ap.add_argument("--decomp", default="auto", choices=["auto", "tracer", "slab"],
                    help="N>1: tracer-index shards + all-gather, or latitude slabs + 2-row halo exchange "
                         "(auto: slabs when every rank gets at least 12 rows, SURVEY.md §8e)")
    ap.add_argument("--exchange", default="auto", choices=["auto", "push", "rccl"],
                    help="N>1: how t(tau+1) moves between ranks -- the library's direct push through hipIpc-mapped windows, "
                         "or RCCL (all-gather / send-recv); auto = push if every rank can map its peers and a trial exchange arrives")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE",
                    help="a diagnostic switch of the library (uvic_gpu_set_option), e.g. close_step=1; not the measured configuration")
    ap.add_argument("--segment", type=int, default=0, metavar="NTSPOS",
                    help="ocean steps per coupling segment: the first step of a segment gets new surface forcing, so its MOBI "
                         "sources are not computed a step ahead (0 = the synthetic forcing is constant, the default)")
    ap.add_argument("--one-slab-of", type=int, default=0, metavar="N",
                    help="diagnosis on one GPU: time only the work of a middle rank of an N-rank latitude-slab run "
                         "(no exchange); the JSON line then describes that rank's share, not the metric")
    a = ap.parse_args()

    import torch
    from uvic29_amd import OPTION_SETS, performance_set, synthetic
    from uvic29_amd.tracer import TracerModel

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the tracer step has no CPU path")
    # rehearsal of the N>1 path on a box with one GPU (tests only): every rank on device 0, gloo instead of RCCL
    rehearse = os.environ.get("UVIC_BENCH_REHEARSAL") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)

    imt, jmt, km = (int(x) for x in a.grid.split("x"))
    cfg = OPTION_SETS[a.cfg] if a.cfg in OPTION_SETS else performance_set(int(a.cfg.replace("perf", "")))
    ocean = synthetic.make_ocean(cfg, imt, jmt, km)
    to, so, c = synthetic.load_eos(km)
    nt, nsrc = cfg.nt, cfg.nsrc
    src = None
    if nsrc and not cfg.ntnpzd:      # passive performance shapes: a given source term
        rng = np.random.default_rng(2029)
        src = np.asfortranarray(rng.standard_normal((imt, km, jmt, nsrc)) * 1e-10 * ocean.topo.tmask[..., None])

    from uvic29_amd.parallel import SlabShard, TracerShard
    # BASELINE.json names the decomposition with the configuration: tracer-index shards on the default grid (configs 3
    # and 4, the north star's primary scheme), latitude slabs with a 2-row halo exchange on the refined grid (config 5)
    decomp, decomp_why = a.decomp, "--decomp"
    if decomp == "auto":
        refined = (imt, jmt, km) == (202, 202, 32)
        decomp = "slab" if world > 1 and refined and (jmt - 2) // world >= 12 else "tracer"
        decomp_why = ("BASELINE config 5 (refined grid): latitude slabs" if decomp == "slab"
                      else "BASELINE configs 3-4: tracer-index shards (--decomp slab for the latitude-slab split)")
    if decomp == "slab" and world > 1:
        shard = SlabShard(jmt, world, rank, exchange=a.exchange)
        shard.nt_model, shard.nt_local = nt, nt
    else:
        decomp = "tracer"
        shard = TracerShard(nt, world, rank, exchange=a.exchange)
    if shard.nt_model != nt:          # pad the tracer dimension with inert tracers (see parallel.py)
        ocean = synthetic.pad_tracers(ocean, shard.nt_model)
    m = TracerModel(imt, jmt, km, shard.nt_model, nsrc, cfg.ntnpzd, device=local_rank)
    # The communicator only now: the library's four streams must take the device's four hardware queues before RCCL's
    # own streams exist (a 12-row slab steps in 0.21 ms this way and in 0.34 ms the other way round).
    if world > 1:
        import torch.distributed as dist
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    m.load_ocean(ocean, to, so, c, src=src)
    if cfg.ntnpzd:
        m.set_mobi(ocean)
    shard.apply(m)
    for kv in a.option:
        if not kv.startswith("iso2="):      # (iso2: a switch of the Python time loop, below)
            m.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    if a.one_slab_of > 1 and world == 1:
        from uvic29_amd.parallel import slab_rows
        js, je = slab_rows(jmt, a.one_slab_of, a.one_slab_of // 2)
        m.set_shard(js=js, je=je)

    from uvic29_amd.tracer import TimeLoop
    loop = TimeLoop(m, ocean.params.dtts, ocean.params.nmix, shard=shard if world > 1 else None, segment=a.segment,
                    iso2=any(kv == "iso2=1" for kv in a.option))

    def one_step():
        loop.step()

    def barrier():
        m.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # The HBM denominator is measured first, on every rank: it also brings the card from its idle power
    # state to its operating clocks, which otherwise takes the first ~60 steps of the loop (tools/step_trace.py: blocks
    # of 8 steps from a cold start run 0.41, 0.34, 0.32, 0.32, 0.31, 0.31, 0.30, ... ms per step); a time loop of hours
    # runs in the steady state, and that is what W warm-up + K timed steps should see whatever W is.
    try:
        hbm = hbm_probe(torch)
    except Exception as e:   # never let the side measurement break the bench line
        hbm = {"error": str(e)}
    for _ in range(a.warmup):
        one_step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        one_step()
    t_submit = time.perf_counter() - t0      # host time to enqueue the K steps
    barrier()
    el = time.perf_counter() - t0
    # the same K steps once more with HIP events around every kernel, each on the stream it is launched
    # on (recording ~17 events per step costs a few per cent, so it is kept out of `value`)
    m.profile_live(True)
    t1 = time.perf_counter()
    for _ in range(a.steps):
        one_step()
    barrier()
    el_instr = time.perf_counter() - t1
    live = m.profile_read()
    # the same loop with the reference's coupling rhythm: ocean segments of 4 steps (run/control.in: segtim 5 d, dtts
    # 1.25 d), whose first step gets new surface forcing and computes its MOBI sources in line
    seg4_ms = None
    if world == 1 and a.segment == 0 and cfg.ntnpzd and a.one_slab_of <= 1:
        loop4 = TimeLoop(m, ocean.params.dtts, ocean.params.nmix, segment=4)
        for _ in range(4):
            loop4.step()
        barrier()
        t4 = time.perf_counter()
        for _ in range(a.steps):
            loop4.step()
        barrier()
        seg4_ms = (time.perf_counter() - t4) / a.steps * 1e3
    if world > 1:
        tt = torch.tensor([el], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())
    units = imt * jmt * km * nt
    value = units * a.steps / el

    out = None
    if rank == 0:
        # restore a sane state (the timed loop may have drifted far) and profile per kernel
        m.load_ocean(ocean, to, so, c, src=src)
        shard.apply(m)
        if a.one_slab_of > 1 and world == 1:
            m.set_shard(js=js, je=je)
        iso = m.profile(nrep=10)   # the same kernels back to back on one stream, nothing overlapped
        # §8(f) rank-1 rows (producers of the shared inputs), timed on their own: not part of `value`
        nxt = {}
        try:
            tid = synthetic.make_tidal(ocean.grid, ocean.topo, ocean.params.kappa_h)
            m.load_velocity(ocean)
            m.load_tidal(ocean, tid)
            m.set_params(diff_cbt_has_k33=1)
            m.profile_live(True)
            for _ in range(10):
                m.adv_vel(); m.isopyc(); m.vmixc()
            pr = m.profile_read()
            nxt = {k: round(pr[k], 5) for k in ("adv_vel_hor", "adv_vel_vert", "vmixc") if k in pr}
            m.set_params(diff_cbt_has_k33=0)
            m.set_filter(ocean, synthetic.make_filter(ocean.grid, km))     # §8(f) rank 3: polar Fourier filter
            pf = m.profile(nrep=3)
            if "filt" in pf:
                nxt["filt"] = round(pf["filt"], 5)
            m.set_filter(ocean, None)
            if world == 1 and a.one_slab_of <= 1 and nt >= 2:    # single-rank runs on the whole grid only
                # §8(f) rank 4: baroclinic momentum step (state + clinic with the polar filter filuv and the sbc accumulation)
                mom = synthetic.make_momentum(ocean.grid, ocean.topo, ocean.u)
                m.load_momentum(ocean, mom)
                m.set_filter_u(ocean, synthetic.make_filter_u(ocean.grid, km))
                m.state(); m.clinic_only(True)
                m.sync()
                tq = time.perf_counter()
                for _ in range(50):
                    m.state_async(); m.clinic_async(True)
                m.sync()
                nxt["state_clinic_ms_per_call"] = round((time.perf_counter() - tq) / 50 * 1e3, 5)
                m.profile_live(True)
                for _ in range(20):
                    m.state_async(); m.clinic_async(True)
                m.sync()
                pr = m.profile_read()
                for k in ("state", "clinic_gradp", "clinic_tend", "clinic_finish", "filuv", "filuv_mean"):
                    if k in pr:
                        nxt[k] = round(pr[k], 5)
                # the whole memory-window loop of mom.F on the device (everything but tropic): tracer step with its look-ahead
                # chains + add_ext_mode, state, adv_vel, clinic; one wait per step (zu back to the host, psi up)
                from uvic29_amd.tracer import OceanLoop
                m.load_ocean(ocean, to, so, c, src=src)
                m.load_velocity(ocean)
                m.load_momentum(ocean, mom)
                oloop = OceanLoop(m, ocean.params.dtts, mom.dtuv)
                psi0 = np.zeros((imt, jmt), order="F")
                for _ in range(8):
                    oloop.step(psi0)
                tq = time.perf_counter()
                for _ in range(40):
                    zu_last = oloop.step(psi0)
                nxt["ocean_loop_ms_per_step"] = round((time.perf_counter() - tq) / 40 * 1e3, 5)
                nxt["ocean_loop_finite"] = bool(np.isfinite(zu_last).all() and np.isfinite(m.download("u1")).all())
                m.set_filter_u(ocean, None)
                if not a.no_cpu_baseline:
                    cms = clinic_cpu_baseline(imt, jmt, km)
                    if cms is not None:
                        nxt["clinic_cpu_reference_ms_per_call"] = round(cms, 3)     # 1 core, compiled reference Fortran
        except Exception as e:   # never let the side measurement break the bench line
            nxt = {"error": str(e)}
        prof = live
        names = [k for k in prof if kernel_alg_bytes(k, nt, nsrc)]
        dom = max(names, key=lambda k: prof[k])
        # tracers of the dominant launch: when T and S take their passes first on the side stream (colx_fct_ts, colx_upd_conv_ts:
        # single rank and latitude slabs), the main-stream launch of colfct / colupd holds the other nt-2
        ts_split = any(k in prof for k in ("colx_fct_ts", "colfct_ts", "fct_rows_ts"))   # T,S in kernels of their own
        nt_launch = shard.nt_local - 2 if ts_split else shard.nt_local
        local_units = imt * jmt * km * nt_launch
        if decomp == "slab":
            local_units = imt * (shard.je - shard.js + 1 + 2) * km * nt_launch     # pass A also does one row beyond each side
        ach = kernel_alg_bytes(dom, nt_launch, nsrc) * local_units / (prof[dom] * 1e-3) / 1e9
        step_gbs = b_alg(nt, nsrc) * value / 1e9
        traffic, traffic_src, valu_insts = pmc_traffic(dom, world == 1 and a.cfg == "c30" and a.grid == "102x102x19", split=ts_split)
        hbm_meas = max(hbm.get("copy_GBs", 0.0), hbm.get("triad_GBs", 0.0)) if hbm else 0.0
        out = {
            "metric": "tracer-cell updates/s (imt*jmt*km*nt)", "value": value, "unit": "cell-updates/s",
            # the metric counts the declared sizes; the same rate over the interior cells only, alongside
            "value_interior_cells": (imt - 2) * (jmt - 2) * km * nt * a.steps / el,
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": el / a.steps * 1e3,
            "ms_per_step_instrumented": el_instr / a.steps * 1e3, "host_submit_ms_per_step": t_submit / a.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            **({"diagnosis": f"one middle slab of {a.one_slab_of}: rows {js}..{je} only, no exchange; not the metric"} if a.one_slab_of > 1 and world == 1 else {}),
            **({"diagnosis_options": a.option} if a.option else {}),
            "config": {"workload": f"{cfg.name} nt={nt} nsrc={nsrc} {imt}x{jmt}x{km}: isopyc + tracer step "
                                   f"(FCT adv_flux, isoflux, explicit update, invtri, convct2"
                                   f"{', MOBI sources (mobi_driver/mobi_src/co2calc_SWS)' if m_has_mobi(m) else ', source term given'})",
                       "grid": a.grid, "nt": nt,
                       "forcing": ("constant: every leapfrog step looks one step ahead" if a.segment == 0 else
                                   f"renewed every {a.segment} steps: the first step of a segment computes its MOBI sources in line"), "parallelism": (f"latitude-slab x{world}, 2-row halo exchange" if decomp == "slab" else f"tracer-shard x{world}"),
                          "decomposition": decomp_why,
                          **({"exchange": ("direct push (hipIpc-mapped windows, library kernels)" if shard.pushing else
                                           "gloo through the host (rehearsal)" if rehearse else "RCCL")} if world > 1 else {})},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         # what actually binds the kernel (fp64 VALU issue, not HBM): wave-level VALU instructions of the
                         # launch (SQ_INSTS_VALU of the same PMC passes, scaled to all waves) x 4 cycles per wave64 fp64
                         # instruction over 1024 SIMDs at 2.4 GHz x the launch's duration in the loop
                         **({"valu_wave_insts": valu_insts,
                             "valu_issue_frac": valu_insts * 4.0 / (1024 * 2.4e9 * prof[dom] * 1e-3)} if valu_insts else {}),
                         **({"peak_measured": hbm_meas, "frac_of_measured": ach / hbm_meas, "hbm_probe": hbm} if hbm_meas else {}),
                         "note": "dominant HBM-side (transport) kernel; kernel_ms are HIP-event means over a second, instrumented "
                                 "pass of the same K steps (ms_per_step_instrumented), where the look-ahead chains (MOBI, isopyc) "
                                 "and the T,S passes run beside the main-stream kernels on the side streams; "
                                 "kernel_ms_isolated: the unfused kernels one after the other on one stream",
                         "kernel_ms": {k: round(v, 5) for k, v in prof.items()},
                         "kernel_ms_isolated": {k: round(v, 5) for k, v in iso.items()},
                         "next_rows_kernel_ms": nxt,
                         "tracers_in_launch": nt_launch,
                         "alg_bytes_per_cell_update": kernel_alg_bytes(dom, nt_launch, nsrc)},
            "step_hbm": {"alg_bytes_per_cell_update": b_alg(nt, nsrc), "achieved_GBs": step_gbs,
                         "frac_of_peak": step_gbs / HBM_PEAK_GBS,
                         **({"frac_of_measured": step_gbs / hbm_meas} if hbm_meas else {})},
        }
        if seg4_ms is not None:
            out["segment4_ms_per_step"] = seg4_ms
        if world == 1 and not a.no_overlay and a.one_slab_of <= 1:
            try:
                ov = overlay_baseline(ocean)
            except Exception as e:      # never let the side measurement break the bench line
                ov = {"error": str(e)}
            if ov is not None:
                out["overlay"] = ov
                if "loop_ms" in ov:
                    out["overlay_ms_per_step"] = ov["loop_ms"]
            try:
                oo = ocean_overlay_baseline(imt, jmt, km) if a.cfg == "c30" else None
            except Exception as e:
                oo = {"error": str(e)}
            if oo is not None:
                out["overlay_ocean_loop"] = oo
        if not a.no_cpu_baseline and world == 1:      # the CPU baseline is reported on rank 0 at N=1 only
            out["cpu_baseline"] = cpu_baseline(ocean, to, so, c, src)
            try:
                out["cpu_baseline_ncore"] = cpu_baseline_ncore(ocean, to, so, c, src)
            except Exception as e:      # a courtesy figure: never let it break the bench line
                out["cpu_baseline_ncore"] = {"error": str(e)}
        print(json.dumps(out), flush=True)
    m.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def m_has_mobi(m):
    return bool(getattr(m, "has_mobi", False))


if __name__ == "__main__":
    main()
