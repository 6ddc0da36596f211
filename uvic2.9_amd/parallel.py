"""Multi-GPU decomposition of the tracer step: one process per GPU, RCCL over xGMI.

Tracer-index sharding (BASELINE configs 3-4).  The per-tracer
loop of `tracer` (/root/reference/updates/09/source/mom/tracer.F:902-1167) has
no dependence between tracers except through T,S-derived coefficients, so rank
r transports the contiguous slice of tracers it owns.  Two things couple
tracers inside a column: MOBI needs every biological tracer at tau-1 and
`convct2` (tracer.F:1198) mixes all tracers over ranges found from T and S.
Hence the schedule per step:

    every rank : isopyc (T,S only), MOBI sources        -- replicated, cheap/compute-bound
    rank r     : transport of its tracer slice
    all ranks  : all-gather of t(:,:,:,slice,tau+1)      -- the one real exchange
    every rank : convct2 on all tracers                  -- replicated, ~0.2 % of the step

The tracer dimension is padded to a multiple of the world size so that every
rank contributes an equal, contiguous chunk and the all-gather runs in place on
the device buffer of t(tau+1) (torch.distributed.all_gather_into_tensor, RCCL).
"""
from __future__ import annotations

import numpy as np


def padded_nt(nt: int, world: int) -> int:
    return ((nt + world - 1) // world) * world


def slice_of(nt: int, world: int, rank: int):
    """(n0, nt_local, chunk): first tracer (0-based), number of REAL tracers of
    the rank and the padded chunk length."""
    chunk = padded_nt(nt, world) // world
    n0 = rank * chunk
    return n0, max(0, min(nt, n0 + chunk) - n0), chunk


class _DevArray:
    """Expose a raw device pointer to torch through __cuda_array_interface__."""

    def __init__(self, ptr: int, nelem: int):
        self.__cuda_array_interface__ = {"shape": (nelem,), "typestr": "<f8", "data": (ptr, False), "version": 3,
                                         "strides": None}


class TracerShard:
    def __init__(self, nt: int, world: int = 1, rank: int = 0):
        self.nt, self.world, self.rank = nt, world, rank
        self.nt_model = padded_nt(nt, world) if world > 1 else nt
        self.n0, self.nt_local, self.chunk = slice_of(nt, world, rank)
        self._views = {}
        self._stream = None

    def apply(self, model):
        model.set_shard(n0=self.n0, nt_local=self.nt_local)

    def _tensor(self, model, name):
        import torch
        ptr = model.devptr(name)
        if ptr not in self._views:
            n = model.imt * model.km * model.jmt * model.nt
            self._views[ptr] = torch.as_tensor(_DevArray(ptr, n), device=f"cuda:{model.device}")
        return self._views[ptr]

    def step(self, model):
        """One device-resident step including the exchange (no host sync)."""
        if self.world == 1:
            model.step_async()
            return
        import torch
        import torch.distributed as dist
        from .capi import check
        if self._stream is None:
            self._stream = torch.cuda.ExternalStream(model.lib.uvic_gpu_stream(model.h), device=f"cuda:{model.device}")
        check(model.lib.uvic_gpu_step_pre_async(model.h), "step_pre_async")
        full = self._tensor(model, "t_taup1")
        per = full.numel() // self.world
        mine = full[self.rank * per:(self.rank + 1) * per]
        with torch.cuda.stream(self._stream):
            if dist.get_backend() == "nccl":       # RCCL, in place on the device buffer, on the library's stream
                dist.all_gather_into_tensor(full, mine)
            else:
                # rehearsal backend (gloo has no device all-gather): staged through the host.  Used by
                # tests/test_gpu_multirank.py to run the N>1 schedule with several ranks on ONE GPU.
                self._stream.synchronize()
                host = torch.empty(full.numel(), dtype=full.dtype)
                dist.all_gather_into_tensor(host, mine.cpu())
                full.copy_(host)
        check(model.lib.uvic_gpu_convect_async(model.h), "convect_async")
