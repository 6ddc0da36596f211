"""The N>1 schedule (tracer-index shards, exchange of t(tau+1), replicated convection, MOBI one
step ahead) with two ranks sharing ONE GPU: gloo stands in for RCCL (which refuses two ranks on
one device), everything else is the path bench.py --gpus N runs.  Result after several steps,
including a mixing step, must equal the unsharded run bit for bit."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
pytestmark = pytest.mark.gpu
NSTEP = 5


def _run(cfg_name, imt, jmt, km, world, rank, decomp="tracer", exchange="rccl"):
    from uvic29_amd import OPTION_SETS, performance_set, synthetic
    from uvic29_amd.parallel import SlabShard, TracerShard
    from uvic29_amd.tracer import TimeLoop, TracerModel
    cfg = OPTION_SETS[cfg_name] if cfg_name in OPTION_SETS else performance_set(int(cfg_name.replace("perf", "")))
    ocean = synthetic.make_ocean(cfg, imt, jmt, km)
    to, so, c = synthetic.load_eos(km)
    src = None
    if cfg.nsrc and not cfg.ntnpzd:      # passive performance shape (BASELINE configs 2-3): a given source term
        rng = np.random.default_rng(2029)
        src = np.asfortranarray(rng.standard_normal((imt, km, jmt, cfg.nsrc)) * 1e-10 * ocean.topo.tmask[..., None])
    if decomp == "slab":
        shard = SlabShard(jmt, world, rank, exchange=exchange)
        shard.nt_model = cfg.nt
    else:
        shard = TracerShard(cfg.nt, world, rank, exchange=exchange)
    if shard.nt_model != cfg.nt:
        ocean = synthetic.pad_tracers(ocean, shard.nt_model)
    m = TracerModel(imt, jmt, km, shard.nt_model, cfg.nsrc, cfg.ntnpzd, device=0)
    m.load_ocean(ocean, to, so, c, src=src)
    if cfg.ntnpzd:
        m.set_mobi(ocean)
    m.set_filter(ocean, synthetic.make_filter(ocean.grid, km))      # polar filter on: it must commute with both decompositions
    shard.apply(m)
    loop = TimeLoop(m, ocean.params.dtts, nmix=3, shard=shard if world > 1 else None)
    for _ in range(NSTEP):
        loop.step()
    m.sync()
    out = m.download("t_tau")[..., :cfg.nt].copy()
    m.close()
    return out


def _worker(rank, world, port, out_path, decomp="tracer", grid=(14, 14, 6), cfg="c30", exchange="rccl"):
    import torch.distributed as dist
    for p in (ROOT,):
        sys.path.insert(0, str(p))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    got = _run(cfg, *grid, world, rank, decomp, exchange)
    np.save(f"{out_path}.{rank}.npy", got)
    if rank == 0:
        np.save(f"{out_path}.single.npy", _run(cfg, *grid, 1, 0))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_equal_single_rank(tmp_path):
    import torch.multiprocessing as mp
    out = str(tmp_path / "mr")
    mp.spawn(_worker, args=(2, 29533, out), nprocs=2, join=True)
    ref = np.load(f"{out}.single.npy")
    assert np.isfinite(ref).all()
    for r in range(2):
        got = np.load(f"{out}.{r}.npy")
        assert np.array_equal(got[:, :, 1:-1], ref[:, :, 1:-1]), f"rank {r}"


def test_two_latitude_slabs_on_one_gpu_equal_single_rank(tmp_path):
    """j-slab decomposition (BASELINE config 5): each rank computes its rows, exchanges a 2-row halo of
    t(tau+1) per step; the owned rows equal the single-rank run bit for bit."""
    import torch.multiprocessing as mp
    from uvic29_amd.parallel import slab_rows
    out = str(tmp_path / "slab")
    mp.spawn(_worker, args=(2, 29537, out, "slab"), nprocs=2, join=True)
    ref = np.load(f"{out}.single.npy")
    for r in range(2):
        js, je = slab_rows(14, 2, r)
        got = np.load(f"{out}.{r}.npy")
        assert np.array_equal(got[:, :, js - 1:je], ref[:, :, js - 1:je]), f"rank {r}"


@pytest.mark.parametrize("cfg", ["s37", "f18"])
def test_two_slabs_and_two_shards_of_the_other_mobi_option_sets(tmp_path, cfg):
    """The shipped nt=37 set and set F (general MOBI kernel, uvic_gpu_set_mobi_opt) under both decompositions: MOBI runs
    on the slab's columns only / replicated, with the look-ahead chains, polar filter and mixing steps of the other tests."""
    import torch.multiprocessing as mp
    from uvic29_amd.parallel import slab_rows
    out = str(tmp_path / "sets_slab")
    mp.spawn(_worker, args=(2, 29611, out, "slab", (14, 14, 6), cfg), nprocs=2, join=True)
    ref = np.load(f"{out}.single.npy")
    assert np.isfinite(ref).all()
    for r in range(2):
        js, je = slab_rows(14, 2, r)
        assert np.array_equal(np.load(f"{out}.{r}.npy")[:, :, js - 1:je], ref[:, :, js - 1:je]), f"slab rank {r}"
    out = str(tmp_path / "sets_shard")
    mp.spawn(_worker, args=(2, 29613, out, "tracer", (14, 14, 6), cfg), nprocs=2, join=True)
    ref = np.load(f"{out}.single.npy")
    for r in range(2):
        assert np.array_equal(np.load(f"{out}.{r}.npy")[:, :, 1:-1], ref[:, :, 1:-1]), f"shard rank {r}"


def test_four_latitude_slabs_full_grid(tmp_path):
    """The same on the 102x102x19 grid with four slabs of 25 rows (interior ranks exchange both ways)."""
    import torch.multiprocessing as mp
    from uvic29_amd.parallel import slab_rows
    out = str(tmp_path / "slab4")
    mp.spawn(_worker, args=(4, 29541, out, "slab", (102, 102, 19)), nprocs=4, join=True)
    ref = np.load(f"{out}.single.npy")
    for r in range(4):
        js, je = slab_rows(102, 4, r)
        got = np.load(f"{out}.{r}.npy")
        assert np.array_equal(got[:, :, js - 1:je], ref[:, :, js - 1:je]), f"rank {r}"


@pytest.mark.parametrize("world", [2, 4])
def test_config3_tracer_shards_nt15_full_grid(tmp_path, world):
    """BASELINE config 3: nt = 15 on 100x100x19, tracer-index shards over 2 and 4 ranks (the tracer dimension is padded
    to a multiple of the world size), all-gather of t(tau+1) per step, replicated convection and polar filter: every
    rank ends with the single-rank result bit for bit."""
    import torch.multiprocessing as mp
    out = str(tmp_path / f"ts{world}")
    mp.spawn(_worker, args=(world, 29581 + world, out, "tracer", (102, 102, 19), "perf15"), nprocs=world, join=True)
    ref = np.load(f"{out}.single.npy")
    assert np.isfinite(ref).all() and ref.shape[3] == 15
    for r in range(world):
        got = np.load(f"{out}.{r}.npy")
        assert np.array_equal(got[:, :, 1:-1], ref[:, :, 1:-1]), f"rank {r}"


def test_config4_tracer_shards_nt30_mobi_full_grid(tmp_path):
    """BASELINE config 4 as worded -- the nt = 30 MOBI-isotope set on 100x100x19 under tracer-index shards -- with as many
    ranks as one GPU box admits beside the test process (four; the eight-rank split of the same schedule is rehearsed on
    the CPU, tests/test_parallel_gloo.py): T and S (bit-exact kernels) on rank 0, MOBI and isopyc replicated, the padded
    all-gather, replicated convection and filter; every rank ends with the single-rank result bit for bit."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "ts4c30")
    mp.spawn(_worker, args=(4, 29621, out, "tracer", (102, 102, 19), "c30"), nprocs=4, join=True)
    ref = np.load(f"{out}.single.npy")
    assert np.isfinite(ref).all() and ref.shape[3] == 30
    for r in range(4):
        got = np.load(f"{out}.{r}.npy")
        assert np.array_equal(got[:, :, 1:-1], ref[:, :, 1:-1]), f"rank {r}"


def test_config5_four_slabs_refined_grid_nt30(tmp_path):
    """BASELINE config 5 as worded: the refined 200x200x32 grid with nt = 30 and MOBI, latitude slabs (four here, ranks on
    one GPU) with the 2-row halo exchange: owned rows equal the single-rank run bit for bit over five steps."""
    import torch.multiprocessing as mp
    from uvic29_amd.parallel import slab_rows
    out = str(tmp_path / "slab4r")
    mp.spawn(_worker, args=(4, 29591, out, "slab", (202, 202, 32)), nprocs=4, join=True)
    ref = np.load(f"{out}.single.npy")
    assert np.isfinite(ref).all()
    for r in range(4):
        js, je = slab_rows(202, 4, r)
        got = np.load(f"{out}.{r}.npy")
        assert np.array_equal(got[:, :, js - 1:je], ref[:, :, js - 1:je]), f"rank {r}"


@pytest.mark.parametrize("decomp,world,grid", [("tracer", 2, (14, 14, 6)), ("slab", 2, (14, 14, 6)), ("slab", 4, (102, 102, 19)),
                                               ("tracer", 4, (102, 102, 19)),
                                               # rank counts that divide neither the rows nor the tracers evenly
                                               ("slab", 3, (102, 102, 19)), ("tracer", 3, (14, 14, 6)), ("tracer", 5, (102, 102, 19)),
                                               ("slab", 6, (102, 102, 19))])
def test_direct_push_between_processes_on_one_gpu(tmp_path, decomp, world, grid):
    """The library's own exchange (uvic_gpu_push_*): each rank's pack kernel writes straight into the receive window its
    peer exported through hipIpc, raises the peer's arrival counter, and waits on the device for its own -- no
    torch.distributed call on the data path (gloo carries the 128-byte handles once).  Separate processes on one GPU
    stand in for the GPUs of a node; owned rows / all tracers equal the single-rank run bit for bit over five steps,
    a mixing step and the look-ahead chains included."""
    import torch.multiprocessing as mp
    from uvic29_amd.parallel import slab_rows
    out = str(tmp_path / "push")
    port = 29641 + 2 * world + (decomp == "slab")
    mp.spawn(_worker, args=(world, port, out, decomp, grid, "c30", "push"), nprocs=world, join=True)
    ref = np.load(f"{out}.single.npy")
    assert np.isfinite(ref).all()
    for r in range(world):
        got = np.load(f"{out}.{r}.npy")
        if decomp == "slab":
            js, je = slab_rows(grid[1], world, r)
            assert np.array_equal(got[:, :, js - 1:je], ref[:, :, js - 1:je]), f"rank {r}"
        else:
            assert np.array_equal(got[:, :, 1:-1], ref[:, :, 1:-1]), f"rank {r}"


def test_direct_push_reports_a_peer_that_never_arrives():
    """The wait for a peer is a kernel with a time limit: a slab whose northern neighbour never pushes ends with an error at
    the next sync (naming the side), not with a hung queue."""
    import time
    from uvic29_amd import OPTION_SETS, synthetic
    from uvic29_amd.capi import UvicGpuError
    from uvic29_amd.tracer import TracerModel
    cfg = OPTION_SETS["c30"]
    ocean = synthetic.make_ocean(cfg, 14, 14, 6)
    to, so, c = synthetic.load_eos(6)
    m = TracerModel(14, 14, 6, cfg.nt, cfg.nsrc, cfg.ntnpzd, device=0)
    m.load_ocean(ocean, to, so, c)
    m.set_shard(js=4, je=9)
    assert m.lib.uvic_gpu_push_setup(m.h, 1, 0, 2) == 0, m.last_error()
    assert m.lib.uvic_gpu_push_open(m.h, 0, None) == 0, m.last_error()
    m.set_option("push_wait_ms", 50)
    t0 = time.perf_counter()
    assert m.lib.uvic_gpu_push_exchange(m.h, -1, 0) == 0, m.last_error()    # pushes north (to itself, as "from the south"), waits for the north
    with pytest.raises(UvicGpuError, match="north"):
        m.sync()
    assert time.perf_counter() - t0 < 5.0
    m.sync()                      # the error is reported once
    m.close()


def _rccl_loop_worker(rank, port, out_path, backend):
    """TimeLoop over a slab whose two neighbours are the rank itself: the asynchronous RCCL send/recv of the halo rows,
    ordered against packing, unpacking and the four-stream look-ahead schedule by the stream alone (backend "nccl"), or
    the host-staged rehearsal path (backend "gloo") that synchronises every step.  Same bits expected."""
    import torch
    import torch.distributed as dist
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    from uvic29_amd import OPTION_SETS, synthetic
    from uvic29_amd.parallel import SlabShard
    from uvic29_amd.tracer import TimeLoop, TracerModel
    cfg = OPTION_SETS["c30"]
    ocean = synthetic.make_ocean(cfg, 14, 14, 6)
    to, so, c = synthetic.load_eos(6)
    m = TracerModel(14, 14, 6, cfg.nt, cfg.nsrc, cfg.ntnpzd, device=0)        # the model before the communicator
    if backend == "nccl":
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    m.load_ocean(ocean, to, so, c)
    m.set_mobi(ocean)
    sl = SlabShard(14, 1, 0)
    sl.js, sl.je = 5, 8
    sl.nt_model = cfg.nt
    sl.apply(m)
    sl.world = 2                      # so that after_step exchanges; the peers are overridden below
    if backend == "nccl":
        orig = sl.exchange
        sl.exchange = lambda model, name="t_taup1", peers=None: orig(model, name, peers=(0, 0))
    else:
        # the comparison run: the same self-exchange staged through the host with a full synchronisation around it
        # (gloo cannot send to the sending rank itself)
        from uvic29_amd.capi import check

        def host_exchange(model, name="t_taup1", peers=None):
            send_s, send_n, recv_s, recv_n = sl._staging(model)
            check(model.lib.uvic_gpu_halo_pack(model.h, 1, 1), "halo_pack")
            model.sync(); torch.cuda.synchronize()
            hs, hn = send_s.cpu(), send_n.cpu()
            recv_s.copy_(hs); recv_n.copy_(hn)
            torch.cuda.synchronize()
            check(model.lib.uvic_gpu_halo_unpack(model.h, 1, 1), "halo_unpack")
        sl.exchange = host_exchange
    loop = TimeLoop(m, ocean.params.dtts, nmix=3, shard=sl)
    for _ in range(6):                # includes two forward (mixing) steps
        loop.step()
    m.sync()
    np.save(out_path, m.download("t_tau"))
    m.close()
    if backend == "nccl":
        dist.destroy_process_group()


def test_rccl_halo_exchange_inside_the_time_loop(tmp_path):
    import torch.multiprocessing as mp
    outs = {}
    for backend, port in (("nccl", 29555), ("host", 29557)):
        out = str(tmp_path / f"loop_{backend}.npy")
        mp.spawn(_rccl_loop_worker, args=(port, out, backend), nprocs=1, join=True)
        outs[backend] = np.load(out)
    assert np.isfinite(outs["nccl"]).all()
    assert np.array_equal(outs["nccl"][:, :, 4:8], outs["host"][:, :, 4:8])      # the slab's own rows 5..8


def _rccl_worker(rank, port, out_path):
    """The RCCL code paths themselves (the rehearsals above use gloo): one rank, backend "nccl"."""
    import torch
    import torch.distributed as dist
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from uvic29_amd import OPTION_SETS, synthetic
    from uvic29_amd.parallel import HALO, SlabShard, TracerShard
    from uvic29_amd.tracer import TracerModel
    cfg = OPTION_SETS["p2"]
    ocean = synthetic.make_ocean(cfg, 14, 14, 6)
    to, so, c = synthetic.load_eos(6)
    m = TracerModel(14, 14, 6, cfg.nt, cfg.nsrc, cfg.ntnpzd, device=0)
    m.load_ocean(ocean, to, so, c)
    m.set_params(c2dtts=2.0 * ocean.params.dtts)
    m.step_async()
    m.sync()
    before = m.download("t_taup1").copy()
    # in-place all-gather on the library's stream through an ExternalStream: the identity for one rank
    ts = TracerShard(cfg.nt, 1, 0)
    ts.gather(m)
    m.sync()
    ok_gather = np.array_equal(m.download("t_taup1"), before)
    # batched send/recv of the packed halo rows: the slab is rows 5..8 and the rank is its own southern and northern
    # neighbour, so its southern rows 5,6 come back as "received from the south" into rows 3,4 and rows 7,8 into 9,10
    sl = SlabShard(14, 1, 0)
    sl.js, sl.je = 5, 8
    sl.apply(m)
    import torch as _t
    sl._stream = _t.cuda.ExternalStream(m.lib.uvic_gpu_stream(m.h), device="cuda:0")
    with _t.cuda.stream(sl._stream):
        sl.exchange(m, "t_taup1", peers=(0, 0))
    m.sync()
    after = m.download("t_taup1")
    want = before.copy()
    # each rank sends [south block, north block] to (south, north) and receives in the same order from them; being both
    # neighbours itself, what it sent south (rows 5,6) is what arrives "from the south" and likewise for the north
    want[:, :, 2:4] = before[:, :, 4:6]
    want[:, :, 8:10] = before[:, :, 6:8]
    ok_p2p = np.array_equal(after, want)
    np.save(out_path, np.array([ok_gather, ok_p2p]))
    m.close()
    dist.destroy_process_group()


def test_rccl_paths_single_rank(tmp_path):
    """all_gather_into_tensor in place on the device buffer and batch_isend_irecv of row blocks through RCCL itself,
    on the library's own stream (what bench.py --gpus N runs; N > 1 needs N GPUs, so one rank here)."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "rccl.npy")
    mp.spawn(_rccl_worker, args=(29551, out), nprocs=1, join=True)
    ok = np.load(out)
    assert ok[0], "in-place all-gather changed the buffer"
    assert ok[1], "row blocks sent through RCCL did not arrive where expected"


@pytest.mark.parametrize("extra", [(), ("--decomp", "slab", "--exchange", "rccl")])
def test_bench_with_two_ranks_on_one_gpu_prints_one_json_line(extra):
    """bench.py --gpus 2 as the driver launches it (torch.distributed.run, one process per rank), with both ranks on
    the one GPU of this box and gloo standing in for RCCL (UVIC_BENCH_REHEARSAL=1): rank 0 prints exactly one JSON line
    with the contract's keys, the whole-job value, the decomposition and the exchange that ran."""
    import json
    import subprocess
    env = dict(os.environ, UVIC_BENCH_REHEARSAL="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29571", str(ROOT / "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", *extra]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["value"] > 0
    # the decomposition follows the BASELINE configuration (default grid: tracer shards), the exchange is the library's own
    if extra:
        assert "latitude-slab x2" in d["config"]["parallelism"] and d["config"]["exchange"].startswith("gloo")
    else:
        assert "tracer-shard x2" in d["config"]["parallelism"] and "configs 3-4" in d["config"]["decomposition"]
        assert d["config"]["exchange"].startswith("direct push")
    assert "cpu_baseline" not in d            # reported at N = 1 only
