#!/usr/bin/env python3
"""What the halo exchange of a latitude-slab run costs per step, measured on ONE GPU: a middle slab of an N-slab run
whose rank is its own southern and northern neighbour (RCCL send/recv to itself, world_size 1).  The transfer itself is
then a local copy, but the packing, RCCL's own stream and its place in the hardware queues are the real thing.
usage: python tools/exchange_cost.py [N]"""
import os
import sys
import time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29581")
torch.cuda.set_device(0)
MODE = os.environ.get("EXCH_INIT", "eager")      # eager: communicator created at init; lazy: at the first collective; none
if MODE == "eager":
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
elif MODE == "lazy":
    dist.init_process_group("nccl", rank=0, world_size=1)
from uvic29_amd import OPTION_SETS, synthetic  # noqa: E402
from uvic29_amd.parallel import SlabShard, slab_rows  # noqa: E402
from uvic29_amd.tracer import TimeLoop, TracerModel  # noqa: E402

cfg = OPTION_SETS["c30"]
oc = synthetic.make_ocean(cfg, 102, 102, 19)
to, so, c = synthetic.load_eos(19)


class SelfSlab(SlabShard):
    def __init__(self, exchange):
        super().__init__(102, 1, 0)
        self.js, self.je = slab_rows(102, n, (n - 1) // 2)
        self.peers = (0, 0) if n > 2 else (None, 0)      # of two slabs the southern one: a northern neighbour only
        self.do_exchange = exchange

    def after_step(self, model):
        if not self.do_exchange:
            return
        if self._stream is None:
            self._stream = torch.cuda.ExternalStream(model.lib.uvic_gpu_stream(model.h), device="cuda:0")
        with torch.cuda.stream(self._stream):
            self.exchange(model, "t_taup1", peers=self.peers)


first = True
for exchange in ((False,) if MODE == "none" else (False, True)):
    m = TracerModel(102, 102, 19, cfg.nt, cfg.nsrc, cfg.ntnpzd, device=0)
    if MODE == "after" and first:        # the communicator after the library's streams exist
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        first = False
    m.load_ocean(oc, to, so, c)
    m.set_mobi(oc)
    sh = SelfSlab(exchange)
    sh.apply(m)
    loop = TimeLoop(m, oc.params.dtts, oc.params.nmix, shard=sh)
    for _ in range(6):
        loop.step()
    m.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(64):
        loop.step()
    ts = time.perf_counter() - t0
    m.sync(); torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(f"[{MODE}] slab 1/{n}, rows {sh.js}..{sh.je}, exchange {'to itself through RCCL' if exchange else 'off'}: "
          f"{el / 64 * 1e3:.4f} ms per step (host queues a step in {ts / 64 * 1e3:.4f} ms)")
    m.close()
if MODE != "none":
    dist.destroy_process_group()
