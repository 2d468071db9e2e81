"""Thin Python handle over the C ABI (include/gangfit.h) for tests and bench.py.

Everything computational happens in libgangfit.so (HIP kernels); this file only marshals numpy arrays.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np

from . import _native as N


def make_apps(drv, exe, k, flags=None) -> np.ndarray:
    """Pack per-app columns into gf_app records (exec_off is filled by the library / by `with_offsets`)."""
    drv = np.asarray(drv, dtype=np.int64).reshape(-1, 3)
    exe = np.asarray(exe, dtype=np.int64).reshape(-1, 3)
    k = np.asarray(k, dtype=np.int32).reshape(-1)
    apps = np.zeros(len(k), dtype=N.APP_DTYPE)
    apps["drv"], apps["exe"], apps["k"] = drv, exe, k
    if flags is not None:
        apps["flags"] = np.asarray(flags, dtype=np.uint32)
    return apps


def with_offsets(apps: np.ndarray) -> Tuple[np.ndarray, int]:
    """Fill exec_off = exclusive prefix sum of k (what gf_fit_batch does internally); returns (apps, total_k)."""
    apps = apps.copy()
    k = np.clip(apps["k"].astype(np.int64), 0, None).astype(np.uint64)  # negative k is rejected by the library
    off = np.zeros(len(apps), dtype=np.uint64)
    if len(apps) > 1:
        off[1:] = np.cumsum(k[:-1])
    apps["exec_off"] = off
    return apps, int(k.sum())


@dataclass
class BatchOut:
    results: np.ndarray  # RESULT_DTYPE
    exec_off: np.ndarray  # uint64
    exec_nodes: np.ndarray  # uint32, concatenated
    failed_at: int = -1

    def placement(self, a: int):
        r = self.results[a]
        n = int(r["exec_len"])
        o = int(self.exec_off[a])
        return bool(r["has_capacity"]), int(r["driver_node"]), self.exec_nodes[o:o + n]


class Context:
    """gf_ctx wrapper. Raises GangfitError on any negative return code — no silent fallback."""

    def __init__(self, device: int = 0, devices=None, options=None):
        """devices: a list of device ids makes ONE context over several devices (gf_init with n_dev > 1: independent batches
        of the plain packers are node-range sharded across them inside the library; an id may repeat).
        options: {key: int} for gf_set_option (test switches: "fifo_generic", "lds_budget", "chain_cache", ...)."""
        self._lib = N.load()
        h = C.c_void_p()
        devs = [device] if devices is None else [int(d) for d in devices]
        ids = (C.c_int * len(devs))(*devs)
        rc = self._lib.gf_init(ids, len(devs), C.byref(h))
        if rc != N.GF_OK:
            raise N.GangfitError(rc, "gf_init failed (no gfx950 device visible?)")
        self._h = h
        self.n_nodes = 0
        for key, value in (options or {}).items():
            self.set_option(key, value)

    def view(self) -> "Context":
        """gf_ctx_view: a context with its own stream and working tables that fits on THIS context's installed snapshot."""
        h = C.c_void_p()
        self._check(self._lib.gf_ctx_view(self._h, C.byref(h)))
        v = _View.__new__(_View)
        v._lib, v._h, v._parent = self._lib, h, self
        return v

    def set_option(self, key: str, value: int):
        self._check(self._lib.gf_set_option(self._h, key.encode(), int(value)))

    def shard_count(self) -> int:
        return int(self._lib.gf_shard_count(self._h))

    def last_error(self) -> str:
        msg = self._lib.gf_last_error(self._h)
        return msg.decode() if msg else ""

    def generation(self):
        """(snapshot epoch, cluster generation, usage generation)"""
        out = np.zeros(3, dtype=np.uint64)
        self._check(self._lib.gf_generation(self._h, N.ptr(out)))
        return tuple(int(v) for v in out)

    def chain_cache_stats(self, reset: bool = False):
        """(chains with the cache armed, chains resumed from a checkpoint, applications evaluated, applications skipped)"""
        out = np.zeros(4, dtype=np.uint64)
        self._check(self._lib.gf_chain_cache_stats(self._h, 1 if reset else 0, N.ptr(out)))
        return tuple(int(v) for v in out)

    # -- lifecycle
    def close(self):
        if getattr(self, "_h", None):
            self._lib.gf_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc != N.GF_OK:
            msg = self._lib.gf_last_error(self._h)
            raise N.GangfitError(rc, msg.decode() if msg else "")

    # -- inputs
    def set_snapshot(self, avail, sched=None):
        """avail/sched: (n_nodes, 3) int64 [cpu milli, mem bytes, gpu]."""
        avail = np.ascontiguousarray(avail, dtype=np.int64).reshape(-1, 3)
        cols = [np.ascontiguousarray(avail[:, j]) for j in range(3)]
        scols = [None, None, None]
        if sched is not None:
            sched = np.ascontiguousarray(sched, dtype=np.int64).reshape(-1, 3)
            scols = [np.ascontiguousarray(sched[:, j]) for j in range(3)]
        self._check(self._lib.gf_snapshot_set(self._h, len(avail), *[N.ptr(c) for c in cols], *[N.ptr(c) for c in scols]))
        self.n_nodes = len(avail)

    def build_snapshot(self, alloc, node_flags, name_rank, overhead=None, res_node=None, res_req=None, zone=None,
                       n_zones: int = 1, driver_label_rank=None, exec_label_rank=None, want_orders: bool = True):
        """gf_snapshot_build: reservation replay + available/schedulable + priority orders on the device, installed as
        the current snapshot.  Returns (driver_order, exec_order) — (None, None) with want_orders=False (nothing of size
        O(n_nodes) then returns to the host)."""
        alloc = np.ascontiguousarray(alloc, dtype=np.int64).reshape(-1, 3)
        n = len(alloc)
        cols = [np.ascontiguousarray(alloc[:, j]) for j in range(3)]
        ocols = [None] * 3
        if overhead is not None:
            overhead = np.ascontiguousarray(overhead, dtype=np.int64).reshape(-1, 3)
            ocols = [np.ascontiguousarray(overhead[:, j]) for j in range(3)]
        rn = np.zeros(0, dtype=np.uint32) if res_node is None else np.ascontiguousarray(res_node, dtype=np.uint32)
        rr = np.zeros((0, 3), dtype=np.int64) if res_req is None else np.ascontiguousarray(res_req, dtype=np.int64).reshape(-1, 3)
        rcols = [np.ascontiguousarray(rr[:, j]) for j in range(3)]
        flags = np.ascontiguousarray(node_flags, dtype=np.uint32)
        ranks = np.ascontiguousarray(name_rank, dtype=np.uint32)
        z = None if zone is None else np.ascontiguousarray(zone, dtype=np.uint32)
        dl = None if driver_label_rank is None else np.ascontiguousarray(driver_label_rank, dtype=np.uint32)
        el = None if exec_label_rank is None else np.ascontiguousarray(exec_label_rank, dtype=np.uint32)
        if not want_orders:
            self._check(self._lib.gf_snapshot_build(self._h, n, *[N.ptr(c) for c in cols], *[N.ptr(c) for c in ocols], len(rn),
                                                    N.ptr(rn), *[N.ptr(c) for c in rcols], N.ptr(flags), N.ptr(z), n_zones,
                                                    N.ptr(ranks), N.ptr(dl), N.ptr(el), None, None, None, None))
            self.n_nodes = n
            return None, None
        d_out, x_out = np.zeros(n + 1, dtype=np.uint32), np.zeros(n + 1, dtype=np.uint32)
        nd, nx = C.c_uint32(0), C.c_uint32(0)
        self._check(self._lib.gf_snapshot_build(self._h, n, *[N.ptr(c) for c in cols], *[N.ptr(c) for c in ocols], len(rn),
                                                N.ptr(rn), *[N.ptr(c) for c in rcols], N.ptr(flags), N.ptr(z), n_zones,
                                                N.ptr(ranks), N.ptr(dl), N.ptr(el), N.ptr(d_out), C.byref(nd), N.ptr(x_out),
                                                C.byref(nx)))
        self.n_nodes = n
        return d_out[: nd.value].copy(), x_out[: nx.value].copy()

    def set_cluster(self, alloc, node_flags, name_rank, overhead=None, zone=None, n_zones: int = 1):
        """gf_cluster_set: the static columns stay resident; build_snapshot_resident then only moves the reservations."""
        alloc = np.ascontiguousarray(alloc, dtype=np.int64).reshape(-1, 3)
        n = len(alloc)
        cols = [np.ascontiguousarray(alloc[:, j]) for j in range(3)]
        ocols = [None] * 3
        if overhead is not None:
            overhead = np.ascontiguousarray(overhead, dtype=np.int64).reshape(-1, 3)
            ocols = [np.ascontiguousarray(overhead[:, j]) for j in range(3)]
        flags = np.ascontiguousarray(node_flags, dtype=np.uint32)
        ranks = np.ascontiguousarray(name_rank, dtype=np.uint32)
        z = None if zone is None else np.ascontiguousarray(zone, dtype=np.uint32)
        self._check(self._lib.gf_cluster_set(self._h, n, *[N.ptr(c) for c in cols], *[N.ptr(c) for c in ocols], N.ptr(flags),
                                             N.ptr(z), n_zones, N.ptr(ranks)))
        self._cluster_n = n

    def usage_reset(self):
        """gf_usage_reset: the resident usage sums back to zero."""
        self._check(self._lib.gf_usage_reset(self._h))

    def usage_apply(self, res_node, res_req=None, sign: int = 1, res_cols=None):
        """gf_usage_apply: add (sign = +1) or remove (sign = -1) reservation entries to / from the resident usage sums.
        res_req = (n, 3) rows or res_cols = (cpu, mem, gpu) columns."""
        rn = np.ascontiguousarray(res_node, dtype=np.uint32)
        if res_cols is not None:
            rcols = [np.ascontiguousarray(c, dtype=np.int64) for c in res_cols]
        else:
            rr = np.ascontiguousarray(res_req, dtype=np.int64).reshape(-1, 3)
            rcols = [np.ascontiguousarray(rr[:, j]) for j in range(3)]
        self._check(self._lib.gf_usage_apply(self._h, len(rn), N.ptr(rn), *[N.ptr(c) for c in rcols], int(sign)))

    def build_snapshot_resident(self, res_node=None, res_req=None, node_flags=None, driver_label_rank=None,
                                exec_label_rank=None, want_orders: bool = True, res_cols=None, resident_usage: bool = False):
        """gf_snapshot_build_resident.  res_cols = (cpu, mem, gpu) contiguous int64 columns of the reservation entries, for
        callers that keep them that way (the C ABI takes columns; res_req = (n, 3) rows is split here on every call)."""
        n = self._cluster_n
        rn = np.zeros(0, dtype=np.uint32) if res_node is None else np.ascontiguousarray(res_node, dtype=np.uint32)
        if res_cols is not None:
            rcols = [np.ascontiguousarray(c, dtype=np.int64) for c in res_cols]
        else:
            rr = np.zeros((0, 3), dtype=np.int64) if res_req is None else np.ascontiguousarray(res_req, dtype=np.int64).reshape(-1, 3)
            rcols = [np.ascontiguousarray(rr[:, j]) for j in range(3)]
        fl = None if node_flags is None else np.ascontiguousarray(node_flags, dtype=np.uint32)
        dl = None if driver_label_rank is None else np.ascontiguousarray(driver_label_rank, dtype=np.uint32)
        el = None if exec_label_rank is None else np.ascontiguousarray(exec_label_rank, dtype=np.uint32)
        self.n_nodes = n
        n_res = N.GF_RESIDENT_USAGE if resident_usage else len(rn)  # resident_usage: build from the sums of usage_apply
        if not want_orders:
            self._check(self._lib.gf_snapshot_build_resident(self._h, n_res, N.ptr(rn), *[N.ptr(c) for c in rcols], N.ptr(fl),
                                                             N.ptr(dl), N.ptr(el), None, None, None, None))
            return None, None
        d_out, x_out = np.zeros(n + 1, dtype=np.uint32), np.zeros(n + 1, dtype=np.uint32)
        nd, nx = C.c_uint32(0), C.c_uint32(0)
        self._check(self._lib.gf_snapshot_build_resident(self._h, n_res, N.ptr(rn), *[N.ptr(c) for c in rcols], N.ptr(fl),
                                                         N.ptr(dl), N.ptr(el), N.ptr(d_out), C.byref(nd), N.ptr(x_out),
                                                         C.byref(nx)))
        return d_out[: nd.value].copy(), x_out[: nx.value].copy()

    def snapshot(self):
        """(avail, sched) of the installed snapshot, (n_nodes, 3) int64 each."""
        a = np.zeros((self.n_nodes, 3), dtype=np.int64)
        s = np.zeros((self.n_nodes, 3), dtype=np.int64)
        self._check(self._lib.gf_snapshot_get(self._h, N.ptr(a), N.ptr(s)))
        return a, s

    def set_zones(self, zone_of_node):
        """Zone id per node (after set_snapshot, before set_orders)."""
        z = np.ascontiguousarray(zone_of_node, dtype=np.uint32)
        assert len(z) == self.n_nodes
        self._check(self._lib.gf_zones_set(self._h, N.ptr(z)))

    def set_orders(self, driver_order, exec_order):
        d = np.ascontiguousarray(driver_order, dtype=np.uint32)
        x = np.ascontiguousarray(exec_order, dtype=np.uint32)
        self._check(self._lib.gf_orders_set(self._h, N.ptr(d), len(d), N.ptr(x), len(x)))

    # -- decisions (host buffers; includes H2D/D2H)
    def fit_batch(self, mode: int, algo: int, apps: np.ndarray) -> BatchOut:
        apps = np.ascontiguousarray(apps, dtype=N.APP_DTYPE)
        apps_off, total_k = with_offsets(apps)
        res = np.zeros(len(apps), dtype=N.RESULT_DTYPE)
        out = np.zeros(total_k + 1, dtype=np.uint32)
        failed = C.c_int32(-1)
        self._check(self._lib.gf_fit_batch(self._h, mode, algo, len(apps), N.ptr(apps), N.ptr(res), N.ptr(out), total_k,
                                           C.byref(failed)))
        return BatchOut(res, apps_off["exec_off"].copy(), out[:total_k], int(failed.value))

    def fit_feasible(self, algo: int, apps: np.ndarray) -> np.ndarray:
        """gf_fit_feasible: HasCapacity of every application (bool array) — what UnschedulablePodMarker reads."""
        apps = np.ascontiguousarray(apps, dtype=N.APP_DTYPE)
        out = np.zeros(len(apps), dtype=np.uint8)
        self._check(self._lib.gf_fit_feasible(self._h, algo, len(apps), N.ptr(apps), N.ptr(out)))
        return out.astype(bool)

    def spark_binpack(self, algo: int, drv, exe, k: int):
        """One decision in the shape of binpack.SparkBinPackFunction. Returns (has_capacity, driver, exec_nodes)."""
        app = make_apps([drv], [exe], [k])
        res = np.zeros(1, dtype=N.RESULT_DTYPE)
        out = np.zeros(k + 1, dtype=np.uint32)
        self._check(self._lib.gf_spark_binpack(self._h, algo, N.ptr(app), N.ptr(res), N.ptr(out), k))
        n = int(res[0]["exec_len"])
        return bool(res[0]["has_capacity"]), int(res[0]["driver_node"]), out[:n].copy()

    def avg_packing_efficiency(self, algo: int, apps: np.ndarray, out: BatchOut) -> np.ndarray:
        """(A, 4) float64 [CPU, Memory, GPU, Max]: ComputeAvgPackingEfficiency over [driver] ++ executors per result."""
        apps_off, total_k = with_offsets(np.ascontiguousarray(apps, dtype=N.APP_DTYPE))
        res = np.ascontiguousarray(out.results)
        ex = np.ascontiguousarray(out.exec_nodes, dtype=np.uint32)
        avg = np.zeros((len(apps_off), 4), dtype=np.float64)
        self._check(self._lib.gf_avg_packing_efficiency(self._h, algo, len(apps_off), N.ptr(apps_off), N.ptr(res),
                                                        N.ptr(ex) if len(ex) else None, len(ex), N.ptr(avg)))
        return avg

    def packing_efficiencies(self, algo: int, drv, exe, driver_node: int, exec_nodes) -> np.ndarray:
        """(n_nodes, 3) float64 per-node efficiencies of one result (PackingResult.PackingEfficiencies)."""
        app = make_apps([drv], [exe], [len(exec_nodes)])
        res = np.zeros(1, dtype=N.RESULT_DTYPE)
        res[0] = (1, driver_node, len(exec_nodes), 1)
        ex = np.ascontiguousarray(exec_nodes, dtype=np.uint32)
        eff = np.zeros((self.n_nodes, 3), dtype=np.float64)
        self._check(self._lib.gf_packing_efficiencies(self._h, algo, N.ptr(app), N.ptr(res),
                                                      N.ptr(ex) if len(ex) else None, N.ptr(eff)))
        return eff

    def executor_fit(self, exe, reserved=None, minimal_fragmentation: bool = False, hosts=None, node_zone=None,
                     req_zone=None) -> np.ndarray:
        """One node index (GF_NO_NODE = no capacity) per executor request: rescheduleExecutor's first-fit loop, or
        rescheduleExecutorWithMinimalFragmentation.  hosts: (n_req, n_nodes) booleans.  node_zone (n_nodes) + req_zone (n_req,
        GF_ANY_ZONE = anywhere): filterNodesToZone on the device (gf_executor_fit_zoned)."""
        exe = np.ascontiguousarray(exe, dtype=np.int64).reshape(-1, 3)
        r = None if reserved is None else np.ascontiguousarray(reserved, dtype=np.int64).reshape(-1, 3)
        bits = None
        if hosts is not None:
            h = np.asarray(hosts, dtype=bool).reshape(len(exe), self.n_nodes)
            words = (self.n_nodes + 31) // 32
            pad = np.zeros((len(exe), words * 32), dtype=bool)
            pad[:, : self.n_nodes] = h
            bits = np.packbits(pad.reshape(len(exe), words, 32), axis=2, bitorder="little").view("<u4").reshape(len(exe), words)
            bits = np.ascontiguousarray(bits)
        out = np.zeros(len(exe), dtype=np.uint32)
        if node_zone is not None or req_zone is not None:  # the zone step of the executor Filter on the device
            nz = np.ascontiguousarray(node_zone, dtype=np.uint32).reshape(-1)
            qz = np.ascontiguousarray(req_zone, dtype=np.uint32).reshape(-1)
            assert len(nz) == self.n_nodes and len(qz) == len(exe)
            self._check(self._lib.gf_executor_fit_zoned(self._h, int(minimal_fragmentation), len(exe), N.ptr(exe), N.ptr(r),
                                                        N.ptr(bits), N.ptr(nz), N.ptr(qz), N.ptr(out)))
            return out
        self._check(self._lib.gf_executor_fit(self._h, int(minimal_fragmentation), len(exe), N.ptr(exe), N.ptr(r),
                                              N.ptr(bits), N.ptr(out)))
        return out

    def find_nodes(self, exe, k, chained: bool = True, want_adds: bool = True):
        """findNodes (failover.go:412-436) for n_req requests.  Returns (placed, last_node, exec_off, exec_nodes, adds):
        placed / last_node per request, the concatenated partial placements, and the `reserved` map in units of exe."""
        exe = np.ascontiguousarray(exe, dtype=np.int64).reshape(-1, 3)
        k = np.ascontiguousarray(k, dtype=np.int32).reshape(-1)
        assert len(exe) == len(k)
        kk = np.clip(k.astype(np.int64), 0, None)
        off = np.zeros(len(k), dtype=np.uint64)
        if len(k) > 1:
            off[1:] = np.cumsum(kk[:-1]).astype(np.uint64)
        total = int(kk.sum())
        res = np.zeros((len(k), 2), dtype=np.uint32)
        out = np.zeros(total + 1, dtype=np.uint32)
        adds = np.zeros((len(k), self.n_nodes), dtype=np.uint32) if want_adds else None
        self._check(self._lib.gf_find_nodes(self._h, int(chained), len(k), N.ptr(exe), N.ptr(k), N.ptr(res), N.ptr(out), total,
                                            N.ptr(adds)))
        return res[:, 0].copy(), res[:, 1].copy(), off, out[:total], adds

    def residual(self) -> np.ndarray:
        out = np.zeros((self.n_nodes, 3), dtype=np.int64)
        self._check(self._lib.gf_residual_get(self._h, N.ptr(out)))
        return out

    # -- device-resident entry point (pointers are raw device addresses, e.g. torch.Tensor.data_ptr())
    def fit_batch_dev(self, mode: int, algo: int, n_apps: int, d_apps: int, d_results: int, d_exec_nodes: int,
                      exec_nodes_len: int, d_failed_at: int = 0, stream: int = 0):
        self._check(self._lib.gf_fit_batch_dev(self._h, mode, algo, n_apps, C.c_void_p(d_apps), C.c_void_p(d_results),
                                               C.c_void_p(d_exec_nodes), exec_nodes_len,
                                               C.c_void_p(d_failed_at) if d_failed_at else None,
                                               C.c_void_p(stream) if stream else None))

    # -- the resident worker of the independent batch (gf_worker_*)
    def worker_fit(self, algo: int, apps: np.ndarray) -> BatchOut:
        """One blocking independent batch through the resident worker (records and answers in pinned memory, no launch)."""
        apps = np.ascontiguousarray(apps, dtype=N.APP_DTYPE)
        apps_off, total_k = with_offsets(apps)
        res = np.zeros(len(apps), dtype=N.RESULT_DTYPE)
        out = np.zeros(total_k + 1, dtype=np.uint32)
        self._check(self._lib.gf_worker_fit(self._h, algo, len(apps), N.ptr(apps), N.ptr(res), N.ptr(out), total_k))
        return BatchOut(res, apps_off["exec_off"].copy(), out[:total_k], -1)

    def worker_submit_dev(self, algo: int, batches) -> int:
        """batches: iterable of (n_apps, d_apps, d_results, d_exec_nodes, exec_nodes_len[, flags]); returns the first ticket."""
        arr = (N.WorkerBatch * len(batches))()
        for i, b in enumerate(batches):
            arr[i].n_apps, arr[i].d_apps, arr[i].d_results, arr[i].d_exec_nodes, arr[i].exec_nodes_len = b[0], b[1], b[2], b[3], b[4]
            arr[i].flags = b[5] if len(b) > 5 else 0
        first = C.c_uint64(0)
        self._check(self._lib.gf_worker_submit_dev(self._h, algo, len(batches), arr, C.byref(first)))
        return int(first.value)

    def worker_batches(self, batches, leave_after: bool = False):
        """A prepared ctypes array for worker_submit_prepared (keeps the marshalling out of a timed region).  leave_after: the
        last batch carries GF_WORKER_LEAVE_AFTER (a bounded stream: a worker this submit launches leaves by itself)."""
        arr = (N.WorkerBatch * len(batches))()
        for i, b in enumerate(batches):
            arr[i].n_apps, arr[i].d_apps, arr[i].d_results, arr[i].d_exec_nodes, arr[i].exec_nodes_len = b[0], b[1], b[2], b[3], b[4]
            arr[i].flags = b[5] if len(b) > 5 else 0
        if leave_after and len(batches):
            arr[len(batches) - 1].flags |= N.GF_WORKER_LEAVE_AFTER
        return arr

    def worker_submit_prepared(self, algo: int, arr) -> int:
        first = C.c_uint64(0)
        self._check(self._lib.gf_worker_submit_dev(self._h, algo, len(arr), arr, C.byref(first)))
        return int(first.value)

    def worker_wait(self, first_ticket: int, n_tickets: int = 1):
        self._check(self._lib.gf_worker_wait(self._h, first_ticket, n_tickets))

    def worker_stop(self):
        self._check(self._lib.gf_worker_stop(self._h))

    def worker_stats(self):
        out = (C.c_uint64 * 4)()
        self._check(self._lib.gf_worker_stats(self._h, out))
        return {"posted": int(out[0]), "complete": int(out[1]), "launches": int(out[2]), "resident": bool(out[3])}

    def worker_geometry(self):
        """(sets, workgroups per set) of the worker's last launch."""
        out = (C.c_uint32 * 2)()
        self._check(self._lib.gf_worker_geometry(self._h, out))
        return int(out[0]), int(out[1])

    def call_phases(self):
        """Host-clock phases (us) of the last blocking independent gf_fit_batch on the zero-copy path."""
        out = (C.c_double * 5)()
        self._check(self._lib.gf_call_phases(self._h, out))
        return {"stage": out[0], "launch": out[1], "wait": out[2], "copy_out": out[3], "total": out[4]}

    def worker_kernel_time(self):
        """(ms on the device, tickets relayed) of the worker's last finished launch — HIP events on the worker's stream."""
        ms, n = C.c_float(), C.c_uint64()
        self._check(self._lib.gf_worker_kernel_time(self._h, C.byref(ms), C.byref(n)))
        return float(ms.value), int(n.value)

    # -- replayable launch sequences (gf_graph_*)
    def graph_begin(self, stream: int = 0):
        self._check(self._lib.gf_graph_begin(self._h, C.c_void_p(stream) if stream else None))

    def graph_end(self, stream: int = 0) -> int:
        g = C.c_void_p()
        self._check(self._lib.gf_graph_end(self._h, C.c_void_p(stream) if stream else None, C.byref(g)))
        return g.value

    def graph_launch(self, graph: int, stream: int = 0):
        self._check(self._lib.gf_graph_launch(self._h, C.c_void_p(graph), C.c_void_p(stream) if stream else None))

    def graph_destroy(self, graph: int):
        self._lib.gf_graph_destroy(self._h, C.c_void_p(graph))

    def timer_begin(self, stream: int = 0):
        self._check(self._lib.gf_timer_begin(self._h, C.c_void_p(stream) if stream else None))

    def timer_end(self) -> float:
        ms = C.c_float(0.0)
        self._check(self._lib.gf_timer_end(self._h, C.byref(ms)))
        return float(ms.value)

    def scan_stats(self, enable: bool = True, reset: bool = False):
        out = np.zeros(10, dtype=np.uint64)
        self._check(self._lib.gf_scan_stats(self._h, int(enable), int(reset), N.ptr(out)))
        self.last_fifo_clock = (int(out[2]), int(out[3]))  # (shader cycles, 100 MHz ticks) of the last FIFO kernel
        self.last_fifo_phases = [int(v) for v in out[4:10]]  # stage, driver scan, executor scan, slow path, commit
        return int(out[0]), int(out[1])

    def chain_profile(self) -> dict:
        """gf_chain_profile: the rare endings of the last instrumented FIFO chain and where its workgroup ran."""
        out = np.zeros(12, dtype=np.uint64)
        self._check(self._lib.gf_chain_profile(self._h, N.ptr(out)))
        names = ("", "unindexed", "bound", "no_driver", "short")
        return {"rare_count": {names[i]: int(out[i]) for i in range(1, 5)},
                "rare_cycles": {names[i]: int(out[5 + i]) for i in range(1, 5)},
                "prologue_cycles": int(out[5]), "chain_end_cycles": int(out[0]),  # solo chain: from the kernel's start
                "hw_id": int(out[10]), "xcc_id": int(out[11])}

    def hbm_probe(self, nbytes: int = 2 << 30, iters: int = 10):
        """(read-only stream GB/s, copy read + write GB/s) this device delivers on `nbytes` buffers."""
        rd, cp = C.c_double(0.0), C.c_double(0.0)
        self._check(self._lib.gf_hbm_probe(self._h, nbytes, iters, C.byref(rd), C.byref(cp)))
        return float(rd.value), float(cp.value)

    def launch_floor(self, stream: int = 0, iters: int = 200) -> float:
        """Microseconds per back-to-back launch of an empty kernel on `stream`."""
        us = C.c_float(0.0)
        self._check(self._lib.gf_launch_floor(self._h, C.c_void_p(stream) if stream else None, iters, C.byref(us)))
        return float(us.value)

    def selftest(self, seed: int = 1, n_cases: int = 256) -> int:
        bad = C.c_uint32(0)
        self._check(self._lib.gf_selftest(self._h, seed, n_cases, C.byref(bad)))
        return int(bad.value)

    def device_info(self) -> dict:
        info = N.DeviceInfo()
        self._check(self._lib.gf_device_info_get(self._h, C.byref(info)))
        return {"name": info.name.decode(), "arch": info.arch.decode(), "compute_units": info.compute_units,
                "lds_bytes_per_cu": info.lds_bytes_per_cu, "wavefront_size": info.wavefront_size,
                "clock_khz": info.clock_khz, "hbm_bytes": info.hbm_bytes}


class _View(Context):
    """A gf_ctx_view handle: same calls, the node count is the parent's."""

    @property
    def n_nodes(self):
        return self._parent.n_nodes

    @n_nodes.setter
    def n_nodes(self, value):
        raise AttributeError("a view does not install snapshots")
