"""Node-range sharding of an independent gang-fit batch across the GPUs of one box.

One process per GPU.  Every rank holds a gf_ctx with the same snapshot and orders and owns one contiguous range of the
priority order (gf_shard_set); a batch is four device steps (include/gangfit.h, csrc/gangfit_shard.inc) with one small
exchange between consecutive steps:

    partials -> all-gather (16 B/app) -> drivers -> all-gather (16 B/app) -> emit -> all-reduce(SUM) of the placement
    buffer (4 B/executor) -> finish

after which every rank holds the full result, bit-identical to the single-GPU gf_fit_batch.  The exchanges are the only
collectives on the data path; on the GPU box they run over RCCL/xGMI (`torch.distributed`, backend "nccl") on device
tensors, in stream order with the kernels.  The messages are KB-sized, i.e. latency-bound: sharding the node table pays
only when one GPU's scan of the table is slower than ~3 collective latencies (50k+ nodes x 10k apps, BASELINE config 4);
for the headline size the throughput path is app sharding with no collective (bench.py).  The FIFO chain does not shard
(each commit must be visible to the next scan): it runs as replicas.

This file is orchestration only: no arithmetic on placements happens in Python, and there is no CPU fallback —
`HipShardEngine` raises when libgangfit or the GPU is missing.  `Comm`/engine are small interfaces so that the N > 1 control
flow is also exercised on CPU (tests/test_sharded_cpu.py: world_size-2 gloo with a numpy engine from tests/).
"""
from __future__ import annotations

import ctypes as C
import threading
from typing import List, Optional

import numpy as np

from . import _native as N
from .context import BatchOut, Context, with_offsets


# ------------------------------------------------------------------------------------------------ communicators
class Comm:
    """What the orchestrator needs from a process group."""

    rank: int = 0
    world: int = 1

    def all_gather(self, t):  # -> tensor [world, *t.shape], same device as t
        raise NotImplementedError

    def all_reduce_sum_(self, t):  # in place
        raise NotImplementedError


class SingleComm(Comm):
    """world_size 1: the exchanges degenerate to views."""

    def all_gather(self, t):
        return t.unsqueeze(0)

    def all_reduce_sum_(self, t):
        return t


class TorchComm(Comm):
    """torch.distributed process group.  "nccl" (= RCCL on ROCm) moves device tensors over xGMI; with a "gloo" group
    device tensors are staged through the host (CPU tests, and two processes sharing one GPU)."""

    def __init__(self, group=None):
        import torch.distributed as dist

        self._dist = dist
        self._group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self._host_staged = dist.get_backend(group) != "nccl"

    def all_gather(self, t):
        import torch

        src = t.contiguous()
        stacked = (self.world,) + tuple(src.shape)
        flat = (self.world * src.shape[0],) + tuple(src.shape[1:])  # concatenated form: accepted by nccl and gloo
        if self._host_staged and src.is_cuda:
            out = torch.empty(flat, dtype=src.dtype)
            self._dist.all_gather_into_tensor(out, src.cpu(), group=self._group)
            return out.view(stacked).to(t.device)
        out = torch.empty(flat, dtype=src.dtype, device=src.device)
        self._dist.all_gather_into_tensor(out, src, group=self._group)
        return out.view(stacked)

    def all_reduce_sum_(self, t):
        if self._host_staged and t.is_cuda:
            h = t.cpu()
            self._dist.all_reduce(h, group=self._group)
            t.copy_(h)
            return t
        self._dist.all_reduce(t, group=self._group)
        return t


class ThreadGroup:
    """In-process stand-in for a process group: `world` threads, one shard each (one GPU, several gf_ctx).  Used by the
    single-GPU parity tests of the sharded path."""

    def __init__(self, world: int):
        self.world = world
        self._barrier = threading.Barrier(world)
        self._slots: List[Optional[object]] = [None] * world

    def comm(self, rank: int) -> "ThreadComm":
        return ThreadComm(self, rank)


class ThreadComm(Comm):
    def __init__(self, group: ThreadGroup, rank: int):
        self._g = group
        self.rank = rank
        self.world = group.world

    def _exchange(self, t):
        import torch

        g = self._g
        if t.is_cuda:
            torch.cuda.current_stream().synchronize()
        g._slots[self.rank] = t
        g._barrier.wait()
        parts = list(g._slots)
        g._barrier.wait()
        return parts

    def all_gather(self, t):
        import torch

        return torch.stack(self._exchange(t.contiguous()), dim=0)

    def all_reduce_sum_(self, t):
        import torch

        parts = self._exchange(t.clone())
        t.copy_(torch.stack(parts, dim=0).sum(dim=0))
        return t


# ------------------------------------------------------------------------------------------------ the HIP engine
_first_look = threading.Lock()


class HipShardEngine:
    """The four device steps through the C ABI, on torch device tensors (torch only owns memory and the stream)."""

    def __init__(self, ctx: Context, shard: int, n_shards: int, device):
        import torch

        with _first_look:  # (one thread at a time: shards of one device as a thread group may all get here first)
            have_gpu = torch.cuda.is_available()
        if not have_gpu:
            raise RuntimeError("HipShardEngine needs an MI355X: the HIP path is the only product path")
        self.ctx = ctx
        self.device = torch.device(device)
        self.shard, self.n_shards = shard, n_shards
        ctx._check(ctx._lib.gf_shard_set(ctx._h, shard, n_shards))
        self._torch = torch
        # one explicit stream for kernels, torch copies and the collectives: the legacy default stream has handle 0,
        # which the C ABI reads as "use the context's own stream" — work split over two streams would race
        self.stream = torch.cuda.Stream(self.device)

    def stream_context(self):
        return self._torch.cuda.stream(self.stream)

    def _stream(self):
        return C.c_void_p(self.stream.cuda_stream)

    def upload_apps(self, apps_off: np.ndarray):
        return self._torch.from_numpy(apps_off.view(np.uint8).copy()).to(self.device)

    def partials(self, algo: int, d_apps, n_apps: int):
        out = self._torch.empty((n_apps, 2), dtype=self._torch.int64, device=self.device)
        c = self.ctx
        c._check(c._lib.gf_shard_partials_dev(c._h, algo, n_apps, C.c_void_p(d_apps.data_ptr()),
                                              C.c_void_p(out.data_ptr()), self._stream()))
        return out

    def drivers(self, algo: int, d_apps, n_apps: int, all_part):
        out = self._torch.empty((n_apps, 4), dtype=self._torch.int32, device=self.device)
        c = self.ctx
        c._check(c._lib.gf_shard_drivers_dev(c._h, algo, n_apps, C.c_void_p(d_apps.data_ptr()),
                                             C.c_void_p(all_part.data_ptr()), C.c_void_p(out.data_ptr()), self._stream()))
        return out

    def emit(self, algo: int, d_apps, n_apps: int, all_part, all_drv, half: int):
        res = self._torch.empty(n_apps * 16, dtype=self._torch.uint8, device=self.device)
        exec2 = self._torch.empty(2 * half, dtype=self._torch.int32, device=self.device)
        c = self.ctx
        c._check(c._lib.gf_shard_emit_dev(c._h, algo, n_apps, C.c_void_p(d_apps.data_ptr()),
                                          C.c_void_p(all_part.data_ptr()), C.c_void_p(all_drv.data_ptr()),
                                          C.c_void_p(res.data_ptr()), C.c_void_p(exec2.data_ptr()), half, self._stream()))
        return res, exec2

    def finish(self, algo: int, d_apps, n_apps: int, all_part, all_drv, res, exec2, half: int):
        c = self.ctx
        c._check(c._lib.gf_shard_finish_dev(c._h, algo, n_apps, C.c_void_p(d_apps.data_ptr()),
                                            C.c_void_p(all_part.data_ptr()), C.c_void_p(all_drv.data_ptr()),
                                            C.c_void_p(res.data_ptr()), C.c_void_p(exec2.data_ptr()), half,
                                            self._stream()))


# ------------------------------------------------------------------------------------------------ orchestration
class ShardedBatch:
    """One pending-app table prepared for repeated sharded evaluation (bench.py times `step`)."""

    def __init__(self, engine, comm: Comm, algo: int, apps: np.ndarray):
        if algo not in (N.GF_ALGO_TIGHTLY_PACK, N.GF_ALGO_DISTRIBUTE_EVENLY):
            raise N.GangfitError(N.GF_ERR_UNSUPPORTED, "node-range sharding serves tightly-pack and distribute-evenly only")
        self.engine, self.comm, self.algo = engine, comm, algo
        apps = np.ascontiguousarray(apps, dtype=N.APP_DTYPE)
        self.apps_off, self.total_k = with_offsets(apps)
        self.n_apps = len(apps)
        self.half = self.total_k + 1
        with engine.stream_context():
            self.d_apps = engine.upload_apps(self.apps_off)
        self.res = self.exec2 = None

    def _placements(self):
        """What the all-reduce carries: the placement half of the buffer; the second half (the capacities of pass 1's nodes) is
        only written and read by distribute-evenly (shard_emit_kernel / shard_finish_kernel)."""
        return self.exec2 if self.algo == N.GF_ALGO_DISTRIBUTE_EVENLY else self.exec2[: self.half]

    def step(self):
        e, c, algo, n = self.engine, self.comm, self.algo, self.n_apps
        with e.stream_context():
            part = e.partials(algo, self.d_apps, n)
            all_part = c.all_gather(part)
            drv = e.drivers(algo, self.d_apps, n, all_part)
            all_drv = c.all_gather(drv)
            self.res, self.exec2 = e.emit(algo, self.d_apps, n, all_part, all_drv, self.half)
            c.all_reduce_sum_(self._placements())
            e.finish(algo, self.d_apps, n, all_part, all_drv, self.res, self.exec2, self.half)

    def step_timed(self):
        """step(), with a pair of events on the engine's stream around each of the three exchanges (GPU engines only).  Returns
        [all-gather of the partials, all-gather of the drivers, all-reduce of the placements] in microseconds of device time —
        what the data-path collectives of one batch cost (bench.py: config.node_sharded.exchange_us)."""
        import torch

        e, c, algo, n = self.engine, self.comm, self.algo, self.n_apps
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
        with e.stream_context():
            part = e.partials(algo, self.d_apps, n)
            ev[0].record(e.stream)
            all_part = c.all_gather(part)
            ev[1].record(e.stream)
            drv = e.drivers(algo, self.d_apps, n, all_part)
            ev[2].record(e.stream)
            all_drv = c.all_gather(drv)
            ev[3].record(e.stream)
            self.res, self.exec2 = e.emit(algo, self.d_apps, n, all_part, all_drv, self.half)
            ev[4].record(e.stream)
            c.all_reduce_sum_(self._placements())
            ev[5].record(e.stream)
            e.finish(algo, self.d_apps, n, all_part, all_drv, self.res, self.exec2, self.half)
        e.stream.synchronize()
        return [ev[2 * i].elapsed_time(ev[2 * i + 1]) * 1e3 for i in range(3)]

    def fetch(self) -> BatchOut:
        with self.engine.stream_context():
            res = self.res.cpu().numpy().view(N.RESULT_DTYPE).copy()
            ex = self.exec2[: self.total_k].cpu().numpy().view(np.uint32).copy()
        return BatchOut(res, self.apps_off["exec_off"].copy(), ex, -1)


def sharded_fit(engine, comm: Comm, algo: int, apps: np.ndarray) -> BatchOut:
    """gf_fit_batch(GF_MODE_INDEPENDENT) with the node table sharded by priority-order range over comm.world GPUs."""
    b = ShardedBatch(engine, comm, algo, apps)
    b.step()
    return b.fetch()
