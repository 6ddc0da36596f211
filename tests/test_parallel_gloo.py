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


def _worker(rank, world, port, out_path):
    for p in (ROOT, ROOT / "oracle", ROOT / "tests" / "hostemu"):
        sys.path.insert(0, str(p))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from uvic29_amd import performance_set, synthetic
    from uvic29_amd.parallel import TracerShard
    import emu
    nt = 5                                   # pads to 6 on two ranks
    shard = TracerShard(nt, world, rank)
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


def test_two_rank_tracer_shard_equals_single_rank(tmp_path):
    sys.path.insert(0, str(ROOT / "tests" / "hostemu"))
    from uvic29_amd import performance_set, synthetic
    import emu
    world = 2
    out = str(tmp_path / "shard")
    mp.spawn(_worker, args=(world, 29511, out), nprocs=world, join=True)
    # single-rank reference with the same padded tracer dimension
    oc = synthetic.pad_tracers(synthetic.make_ocean(performance_set(5), 14, 14, 6), 6)
    to, so, c = synthetic.load_eos(6)
    rng = np.random.default_rng(11)
    src = np.asfortranarray(rng.standard_normal((14, 6, 14, oc.cfg.nsrc)) * 1e-9 * oc.topo.tmask[..., None])
    em = emu.EmuOcean(oc, to, so, c, src=src)
    em.isopyc(); em.transport(); em.convect()
    for r in range(world):
        got = np.load(f"{out}.{r}.npy")
        assert np.array_equal(got[:, :, 1:13], em.a["t_taup1"][:, :, 1:13]), r
