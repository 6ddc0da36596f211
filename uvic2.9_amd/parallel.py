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

Latitude-slab decomposition (BASELINE config 5, `SlabShard`): rank r owns the contiguous rows
js..je of every field and every tracer; isopyc, MOBI, transport and convection all run on the slab
(the T,S-derived fields two rows beyond it, redundantly), and after the step the two outermost
owned rows of t(tau+1) on each side go to the neighbour's halo: 2 x imt x km x nt x 8 B per
neighbour and direction, point to point (RCCL send/recv over xGMI).  Longitude is cyclic and stays
on the rank.  Every rank keeps the full-size arrays (global row indices, no re-indexing); only its
slab and halo are ever touched.

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


def open_direct_push(model, world: int, rank: int, mode: int, peers=None) -> bool:
    """Set up the library's direct push (include/uvic_gpu.h uvic_gpu_push_*): every rank exports the hipIpc handles
    of its receive window and arrival counters, the handles go round through torch.distributed (128 bytes per rank,
    once), and each rank maps those of the peers it writes to.  True if EVERY rank succeeded -- the ranks agree, so
    that all of them take the same path afterwards; on False the caller keeps the RCCL exchange."""
    import ctypes
    import torch
    import torch.distributed as dist
    ok, mine = 1, bytes(128)
    if model.lib.uvic_gpu_push_setup(model.h, world, rank, mode) == 0:
        buf = ctypes.create_string_buffer(128)
        if model.lib.uvic_gpu_push_export(model.h, buf) == 0:
            mine = buf.raw
        else:
            ok = 0
    else:
        ok = 0
    everyone = [None] * world
    dist.all_gather_object(everyone, (ok, mine))
    if not all(o for o, _ in everyone):
        return False
    for peer in (range(world) if peers is None else peers):
        if peer == rank and peers is None:
            continue
        if model.lib.uvic_gpu_push_open(model.h, peer, everyone[peer][1]) != 0:
            ok = 0
            break
    agreed = [None] * world
    dist.all_gather_object(agreed, ok)
    if not all(agreed):
        return False
    # one exchange of whatever t(tau+1) holds now (the step's own exchange overwrites it), waited for: a link that maps
    # but does not deliver shows up here, before anything is measured or trusted
    if mode == 2:
        ok = int(model.lib.uvic_gpu_push_exchange(model.h, rank - 1 if rank > 0 else -1, rank + 1 if rank + 1 < world else -1) == 0)
    else:
        ok = int(model.lib.uvic_gpu_push_exchange(model.h, -1, -1) == 0)
    ok = int(ok and model.lib.uvic_gpu_sync(model.h) == 0)
    dist.all_gather_object(agreed, ok)
    return all(agreed)


class TracerShard:
    def __init__(self, nt: int, world: int = 1, rank: int = 0, exchange: str = "rccl"):
        """exchange: "rccl" (torch.distributed collective), "push" (the library's direct push; an error if it cannot
        be set up), "auto" (push if every rank can set it up, else rccl)."""
        self.nt, self.world, self.rank = nt, world, rank
        self.nt_model = padded_nt(nt, world) if world > 1 else nt
        self.n0, self.nt_local, self.chunk = slice_of(nt, world, rank)
        self._views = {}
        self._stream = None
        self.exchange_kind = exchange
        self.pushing = None        # decided at the first exchange

    def _decide(self, model):
        if self.pushing is None:
            self.pushing = False
            if self.exchange_kind in ("push", "auto") and self.world > 1:
                self.pushing = open_direct_push(model, self.world, self.rank, 1)
                if not self.pushing and self.exchange_kind == "push":
                    raise RuntimeError("direct push could not be set up on every rank: " + model.last_error())
        return self.pushing

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
        self.gather(model)
        check(model.lib.uvic_gpu_convect_async(model.h), "convect_async")

    def gather(self, model):
        """All-gather of t(:,:,:,slice,tau+1), in place on the device buffer, on the library's stream."""
        import torch
        import torch.distributed as dist
        from .capi import check
        if self._decide(model):
            check(model.lib.uvic_gpu_push_exchange(model.h, -1, -1), "push_exchange")
            return
        if self._stream is None:
            self._stream = torch.cuda.ExternalStream(model.lib.uvic_gpu_stream(model.h), device=f"cuda:{model.device}")
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


HALO = 2   # rows: FCT needs R+-Y of rows r+-1, each of which needs t of its own r+-1


def slab_rows(jmt: int, world: int, rank: int):
    """Owned rows (1-based, inclusive) of `rank`: the computed rows 2..jmt-1 split into contiguous slabs."""
    n = jmt - 2
    base, rem = divmod(n, world)
    lo = 2 + rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0) - 1
    return lo, hi


class SlabShard:
    """Latitude-slab decomposition with a 2-row halo exchange of t(tau+1) per step."""

    def __init__(self, jmt: int, world: int = 1, rank: int = 0, exchange: str = "rccl"):
        self.jmt, self.world, self.rank = jmt, world, rank
        self.exchange_kind = exchange   # "rccl" | "push" | "auto", as for TracerShard
        self.pushing = None
        self.js, self.je = slab_rows(jmt, world, rank)
        if world > 1 and self.je - self.js + 1 < HALO:
            raise ValueError(f"slab of {self.je - self.js + 1} rows is thinner than the halo ({HALO}): use fewer ranks")
        self.nt_model = None
        self._views = {}
        self._stream = None

    def apply(self, model):
        model.set_shard(js=self.js, je=self.je)

    def _rows(self, model, name):
        """t field as a (nt, jmt, imt*km) strided view of the device buffer."""
        import torch
        ptr = model.devptr(name)
        if ptr not in self._views:
            n = model.imt * model.km * model.jmt * model.nt
            flat = torch.as_tensor(_DevArray(ptr, n), device=f"cuda:{model.device}")
            self._views[ptr] = flat.view(model.nt, model.jmt, model.imt * model.km)
        return self._views[ptr]

    def _staging(self, model):
        """The library's four staging buffers (send south/north, receive south/north) as torch views, made once."""
        import torch
        if not self._views.get("halo"):
            n = int(model.lib.uvic_gpu_halo_elems(model.h))
            bufs = []
            for which in range(4):
                ptr = model.lib.uvic_gpu_halo_buffer(model.h, which)
                if not ptr:
                    raise RuntimeError("uvic_gpu_halo_buffer: no staging buffer")
                bufs.append(torch.as_tensor(_DevArray(ptr, n), device=f"cuda:{model.device}"))
            self._views["halo"] = bufs
        return self._views["halo"]

    def exchange(self, model, name="t_taup1", peers=None):
        """Send the outermost HALO owned rows of t(tau+1) to the neighbours, receive theirs into the halo rows.
        The library packs and unpacks (one kernel per side, on its stream); only the transfer itself goes through
        torch.distributed, so a step costs one batched send/recv and no allocation.  `peers` = (south, north) ranks
        overrides the neighbour pattern (tests: a rank may name itself)."""
        import torch
        import torch.distributed as dist
        from .capi import check
        assert name == "t_taup1"
        south, north = (self.rank - 1, self.rank + 1) if peers is None else peers
        has_s = south is not None and 0 <= south < max(self.world, 1)
        has_n = north is not None and 0 <= north < max(self.world, 1)
        if not (has_s or has_n):
            return
        send_s, send_n, recv_s, recv_n = self._staging(model)
        check(model.lib.uvic_gpu_halo_pack(model.h, int(has_s), int(has_n)), "halo_pack")
        pairs = ([(south, send_s, recv_s)] if has_s else []) + ([(north, send_n, recv_n)] if has_n else [])
        if dist.get_backend() == "nccl":
            key = ("ops", south if has_s else None, north if has_n else None)
            ops = self._views.get(key)
            if ops is None:                         # the same buffers and peers every step: built once
                ops = []
                for peer, sb, rb in pairs:
                    ops.append(dist.P2POp(dist.isend, sb, peer))
                    ops.append(dist.P2POp(dist.irecv, rb, peer))
                self._views[key] = ops
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        else:                                       # rehearsal backend: staged through the host
            torch.cuda.current_stream().synchronize()
            hs = [sb.cpu() for _, sb, _ in pairs]
            hr = [torch.empty_like(b) for b in hs]
            ws = []
            for (peer, _, _), sb, rb in zip(pairs, hs, hr):
                ws.append(dist.isend(sb, peer))
                ws.append(dist.irecv(rb, peer))
            for w in ws:
                w.wait()
            for (_, _, rb), hb in zip(pairs, hr):
                rb.copy_(hb)
        check(model.lib.uvic_gpu_halo_unpack(model.h, int(has_s), int(has_n)), "halo_unpack")

    def step(self, model):
        """One device-resident step of the slab and the halo exchange (no host sync with RCCL)."""
        model.step_async()
        self.after_step(model)

    def after_step(self, model):
        """The halo exchange of t(tau+1), queued behind the step on the library's stream."""
        if self.world == 1:
            return
        import torch
        if self.pushing is None:
            self.pushing = False
            if self.exchange_kind in ("push", "auto"):
                near = [r for r in (self.rank - 1, self.rank + 1) if 0 <= r < self.world]
                self.pushing = open_direct_push(model, self.world, self.rank, 2, peers=near)
                if not self.pushing and self.exchange_kind == "push":
                    raise RuntimeError("direct push could not be set up on every rank: " + model.last_error())
        if self.pushing:
            from .capi import check
            south = self.rank - 1 if self.rank > 0 else -1
            north = self.rank + 1 if self.rank + 1 < self.world else -1
            check(model.lib.uvic_gpu_push_exchange(model.h, south, north), "push_exchange")
            return
        if self._stream is None:
            self._stream = torch.cuda.ExternalStream(model.lib.uvic_gpu_stream(model.h), device=f"cuda:{model.device}")
        with torch.cuda.stream(self._stream):
            self.exchange(model, "t_taup1")
