"""The N>1 schedule of uvic2.9_amd/parallel.py on the CPU: world_size 2, gloo.
Each rank transports its tracer slice with the host-emulated kernels, the slices
of t(tau+1) are all-gathered in place, convection runs replicated -- the result
must equal the unsharded run bit-for-bit on every rank."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parents[1]


def test_slices_cover_all_tracers():
    from uvic29_amd.parallel import padded_nt, slice_of
    for nt in (2, 8, 15, 30, 37):
        for world in (1, 2, 4, 8):
            ntp = padded_nt(nt, world)
            assert ntp % world == 0 and 0 <= ntp - nt < world
            seen = []
            for r in range(world):
                n0, nloc, chunk = slice_of(nt, world, r)
                assert chunk == ntp // world and n0 == r * chunk
                seen += list(range(n0, n0 + nloc))
            assert seen == list(range(nt))


def _worker(rank, world, port, out_path, nt=5):
    for p in (ROOT, ROOT / "oracle", ROOT / "tests" / "hostemu"):
        sys.path.insert(0, str(p))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from uvic29_amd import performance_set, synthetic
    from uvic29_amd.parallel import TracerShard
    import emu
    shard = TracerShard(nt, world, rank)     # (nt = 5 pads to 6 on two ranks, nt = 30 to 32 on eight)
    oc = synthetic.make_ocean(performance_set(nt), 14, 14, 6)
    oc = synthetic.pad_tracers(oc, shard.nt_model)
    to, so, c = synthetic.load_eos(6)
    rng = np.random.default_rng(11)
    src = np.asfortranarray(rng.standard_normal((14, 6, 14, oc.cfg.nsrc)) * 1e-9 * oc.topo.tmask[..., None])
    em = emu.EmuOcean(oc, to, so, c, src=src)
    em.ctx.n0, em.ctx.nt_local = shard.n0, shard.nt_local
    em.isopyc()
    em.transport(nchunk=2, nthreads=32)
    full = torch.from_numpy(em.a["t_taup1"].reshape(-1, order="F"))   # shares memory (F-contiguous)
    per = full.numel() // world
    mine = full[rank * per:(rank + 1) * per].clone()
    dist.all_gather_into_tensor(full, mine)
    em.convect()
    np.save(f"{out_path}.{rank}.npy", em.a["t_taup1"])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,nt,port", [(2, 5, 29511), (8, 30, 29531)])   # (8 x nt = 30: BASELINE config 4 as worded)
def test_tracer_shards_equal_single_rank(tmp_path, world, nt, port):
    sys.path.insert(0, str(ROOT / "tests" / "hostemu"))
    from uvic29_amd import performance_set, synthetic
    from uvic29_amd.parallel import padded_nt
    import emu
    out = str(tmp_path / "shard")
    mp.spawn(_worker, args=(world, port, out, nt), nprocs=world, join=True)
    # single-rank reference with the same padded tracer dimension
    oc = synthetic.pad_tracers(synthetic.make_ocean(performance_set(nt), 14, 14, 6), padded_nt(nt, world))
    to, so, c = synthetic.load_eos(6)
    rng = np.random.default_rng(11)
    src = np.asfortranarray(rng.standard_normal((14, 6, 14, oc.cfg.nsrc)) * 1e-9 * oc.topo.tmask[..., None])
    em = emu.EmuOcean(oc, to, so, c, src=src)
    em.isopyc(); em.transport(); em.convect()
    for r in range(world):
        got = np.load(f"{out}.{r}.npy")
        assert np.array_equal(got[:, :, 1:13], em.a["t_taup1"][:, :, 1:13]), r


# ---- latitude slabs (BASELINE config 5): 2-row halo exchange of t(tau+1) ----------------------
NSTEP_SLAB = 3


def _slab_steps(em, js, je, exchange):
    """NSTEP_SLAB leapfrog steps of rows js..je with the host-emulated kernels."""
    em.ctx.js, em.ctx.je = js, je
    for _ in range(NSTEP_SLAB):
        em.isopyc(); em.transport(nchunk=2, nthreads=32); em.convect()
        exchange(em.a["t_taup1"])
        a = em.a
        a["t_taum1"], a["t_tau"], a["t_taup1"] = a["t_tau"], a["t_taup1"], a["t_taum1"]
        em.rebind()
    return em.a["t_tau"]


def _slab_worker(rank, world, port, out_path, jmt=14):
    for p in (ROOT, ROOT / "oracle", ROOT / "tests" / "hostemu"):
        sys.path.insert(0, str(p))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from uvic29_amd import performance_set, synthetic
    from uvic29_amd.parallel import HALO, slab_rows
    import emu
    oc = synthetic.make_ocean(performance_set(3), 14, jmt, 6)
    to, so, c = synthetic.load_eos(6)
    em = emu.EmuOcean(oc, to, so, c)
    js, je = slab_rows(jmt, world, rank)

    def exchange(tp):                      # tp (imt,km,jmt,nt), F order; rows are axis 2
        ws, bufs = [], []
        for peer, send, recv in ((rank - 1, slice(js - 1, js - 1 + HALO), slice(js - 1 - HALO, js - 1)),
                                 (rank + 1, slice(je - HALO, je), slice(je, je + HALO))):
            if 0 <= peer < world:
                sb = torch.from_numpy(np.ascontiguousarray(tp[:, :, send, :]))
                rb = torch.empty_like(sb)
                ws += [dist.isend(sb, peer), dist.irecv(rb, peer)]
                bufs.append((recv, rb))
        for w in ws:
            w.wait()
        for recv, rb in bufs:
            tp[:, :, recv, :] = rb.numpy()

    got = _slab_steps(em, js, je, exchange)
    np.save(f"{out_path}.{rank}.npy", got)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,jmt,port", [(2, 14, 29517), (8, 98, 29537)])   # (8 slabs of 12 rows: the thinnest legal slab)
def test_latitude_slabs_equal_single_rank(tmp_path, world, jmt, port):
    sys.path.insert(0, str(ROOT / "tests" / "hostemu"))
    from uvic29_amd import performance_set, synthetic
    from uvic29_amd.parallel import slab_rows
    import emu
    out = str(tmp_path / "slab")
    mp.spawn(_slab_worker, args=(world, port, out, jmt), nprocs=world, join=True)
    oc = synthetic.make_ocean(performance_set(3), 14, jmt, 6)
    to, so, c = synthetic.load_eos(6)
    ref = _slab_steps(emu.EmuOcean(oc, to, so, c), 2, jmt - 1, lambda tp: None)
    assert np.isfinite(ref).all()
    for r in range(world):
        js, je = slab_rows(jmt, world, r)
        got = np.load(f"{out}.{r}.npy")
        assert np.array_equal(got[:, :, js - 1:je], ref[:, :, js - 1:je]), r
    assert slab_rows(102, 8, 0) == (2, 14) and slab_rows(102, 8, 7)[1] == 101
