"""ctypes binding of the plain-C CPU oracle (test infrastructure, NOT product code).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgangfit_oracle.so")

ALGO_TIGHTLY_PACK = 0
ALGO_DISTRIBUTE_EVENLY = 1
ALGO_MINIMAL_FRAGMENTATION = 2
ALGO_AZ_AWARE_TIGHTLY_PACK = 3
ALGO_SINGLE_AZ_TIGHTLY_PACK = 4
ALGO_SINGLE_AZ_MINIMAL_FRAGMENTATION = 5
NO_NODE = 0xFFFFFFFF
APP_SKIPPABLE = 1

APP_DTYPE = np.dtype(
    [("drv", "<i8", (3,)), ("exe", "<i8", (3,)), ("k", "<i4"), ("flags", "<u4")], align=True
)
RESULT_DTYPE = np.dtype(
    [("has_capacity", "<i4"), ("driver_node", "<u4"), ("exec_len", "<u4"), ("evaluated", "<u4")], align=True
)
assert APP_DTYPE.itemsize == 56 and RESULT_DTYPE.itemsize == 16


def build(force: bool = False) -> str:
    """Compile oracle/gangfit_oracle.c with gcc (seconds)."""
    deps = [os.path.join(_HERE, f) for f in ("gangfit_oracle.c", "gangfit_oracle_maps.cpp", "gangfit_oracle.h", "Makefile")]
    stale = (
        force
        or not os.path.exists(_LIB_PATH)
        or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(d) for d in deps)
    )
    if stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libgangfit_oracle.so"])
    return _LIB_PATH


_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        p = C.c_void_p
        L.go_spark_binpack.restype = C.c_int
        L.go_spark_binpack.argtypes = [C.c_int, p, C.c_uint32, p, p, C.c_uint32, p, C.c_uint32, p, p]
        L.go_spark_binpack_closed_form.restype = C.c_int
        L.go_spark_binpack_closed_form.argtypes = L.go_spark_binpack.argtypes
        L.go_fit_independent.restype = None
        L.go_fit_independent.argtypes = [C.c_int, C.c_int, p, C.c_uint32, p, C.c_uint32, p, C.c_uint32, p,
                                         C.c_uint32, p, p, p]
        L.go_fit_fifo_chain.restype = C.c_int32
        L.go_fit_fifo_chain.argtypes = L.go_fit_independent.argtypes
        L.go_fit_independent_ex.restype = None
        L.go_fit_independent_ex.argtypes = [C.c_int, C.c_int, p, p, p, C.c_uint32, p, C.c_uint32, p, C.c_uint32, p,
                                            C.c_uint32, p, p, p, p]
        L.go_fit_fifo_chain_ex.restype = C.c_int32
        L.go_fit_fifo_chain_ex.argtypes = [C.c_int, C.c_int, p, p, p, C.c_uint32, p, C.c_uint32, p, C.c_uint32, p,
                                           C.c_uint32, p, p, p]
        L.go_avg_packing_efficiency_list.restype = None
        L.go_avg_packing_efficiency_list.argtypes = [p, p, C.c_uint32, p, C.c_uint32, p, C.c_uint32, C.c_int, p]
        L.go_node_capacity.restype = C.c_int64
        L.go_node_capacity.argtypes = [p, p, p]
        L.go_packing_efficiency.restype = None
        L.go_packing_efficiency.argtypes = [p, p, C.c_uint32, p, C.c_uint32, p, C.c_uint32, p, p]
        L.go_executor_first_fit.restype = C.c_uint32
        L.go_executor_first_fit.argtypes = [p, C.c_uint32, p, p, C.c_uint32]
        L.go_executor_first_fit_reserved.restype = C.c_uint32
        L.go_executor_first_fit_reserved.argtypes = [p, C.c_uint32, p, p, p, C.c_uint32]
        L.go_executor_min_frag.restype = C.c_uint32
        L.go_executor_min_frag.argtypes = [p, C.c_uint32, p, p, p, C.c_uint32, p]
        L.go_fit_fifo_chain_with_efficiencies.restype = C.c_int32
        L.go_fit_fifo_chain_with_efficiencies.argtypes = [C.c_int, p, p, p, C.c_uint32, p, C.c_uint32, p, C.c_uint32, p,
                                                          C.c_uint32, p, p, p]
        L.go_fit_maps.restype = C.c_int32
        L.go_fit_maps.argtypes = [C.c_int, C.c_int, p, p, C.c_uint32, p, C.c_uint32, p, C.c_uint32, p, C.c_uint32, p, p, p]
        L.go_find_nodes_chain.restype = None
        L.go_find_nodes_chain.argtypes = [C.c_int, p, C.c_uint32, p, p, C.c_uint32, p, C.c_uint32, p, p, p, p]
        _lib = L
    return _lib


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def make_apps(drv, exe, k, flags=None) -> np.ndarray:
    """Pack per-app arrays (A x 3, A x 3, A, A) into the go_app record layout."""
    drv = np.asarray(drv, dtype=np.int64).reshape(-1, 3)
    exe = np.asarray(exe, dtype=np.int64).reshape(-1, 3)
    k = np.asarray(k, dtype=np.int32).reshape(-1)
    apps = np.zeros(len(k), dtype=APP_DTYPE)
    apps["drv"], apps["exe"], apps["k"] = drv, exe, k
    if flags is not None:
        apps["flags"] = np.asarray(flags, dtype=np.uint32)
    return apps


def exec_offsets(k: np.ndarray) -> np.ndarray:
    off = np.zeros(len(k), dtype=np.uint64)
    if len(k) > 1:
        off[1:] = np.cumsum(np.asarray(k[:-1], dtype=np.uint64))
    return off


@dataclass
class BatchOut:
    results: np.ndarray  # RESULT_DTYPE
    exec_off: np.ndarray  # uint64
    exec_nodes: np.ndarray  # uint32, concatenated
    failed_at: int = -1
    avail_after: Optional[np.ndarray] = None
    avg_eff: Optional[np.ndarray] = None  # (A, 4) AvgPackingEfficiency over [driver] ++ executors (zeros if infeasible)

    def placement(self, a: int) -> Tuple[bool, int, np.ndarray]:
        r = self.results[a]
        n = int(r["exec_len"])
        o = int(self.exec_off[a])
        return bool(r["has_capacity"]), int(r["driver_node"]), self.exec_nodes[o:o + n]


def _prep(avail, driver_order, exec_order):
    avail = np.ascontiguousarray(avail, dtype=np.int64).reshape(-1, 3)
    d = np.ascontiguousarray(driver_order, dtype=np.uint32)
    x = np.ascontiguousarray(exec_order, dtype=np.uint32)
    return avail, d, x


def _aux(avail, sched, zone):
    if sched is not None:
        sched = np.ascontiguousarray(sched, dtype=np.int64).reshape(-1, 3)
        assert len(sched) == len(avail)
    if zone is not None:
        zone = np.ascontiguousarray(zone, dtype=np.uint32)
        assert len(zone) == len(avail)
    return sched, zone


def fit_independent(algo: int, avail, apps: np.ndarray, driver_order, exec_order, closed_form: bool = False,
                    sched=None, zone=None) -> BatchOut:
    """sched / zone: SchedulableResources (A x 3) and zone ids per node — needed by the single-AZ packers; with sched
    the per-result AvgPackingEfficiency over [driver] ++ executors is returned in BatchOut.avg_eff."""
    avail, d, x = _prep(avail, driver_order, exec_order)
    sched, zone = _aux(avail, sched, zone)
    apps = np.ascontiguousarray(apps)
    res = np.zeros(len(apps), dtype=RESULT_DTYPE)
    off = exec_offsets(apps["k"])
    out = np.zeros(int(apps["k"].astype(np.int64).sum()) + 1, dtype=np.uint32)
    avg = np.zeros((len(apps), 4), dtype=np.float64) if sched is not None else None
    lib().go_fit_independent_ex(algo, int(closed_form), _ptr(avail), _ptr(sched), _ptr(zone), len(avail), _ptr(apps),
                                len(apps), _ptr(d), len(d), _ptr(x), len(x), _ptr(res), _ptr(off), _ptr(out), _ptr(avg))
    return BatchOut(res, off, out[:-1], avg_eff=avg)


def fit_fifo_chain(algo: int, avail, apps: np.ndarray, driver_order, exec_order, closed_form: bool = False,
                   sched=None, zone=None, with_efficiencies: bool = False) -> BatchOut:
    """with_efficiencies: also build the PackingEfficiencies map of every successful pack like the reference's
    SparkBinPack does (binpack.go:77) — same results, the reference's cost shape (needs sched; literal loops)."""
    avail, d, x = _prep(avail, driver_order, exec_order)
    sched, zone = _aux(avail, sched, zone)
    avail = avail.copy()
    apps = np.ascontiguousarray(apps)
    res = np.zeros(len(apps), dtype=RESULT_DTYPE)
    off = exec_offsets(apps["k"])
    out = np.zeros(int(apps["k"].astype(np.int64).sum()) + 1, dtype=np.uint32)
    if with_efficiencies:
        assert sched is not None and not closed_form
        failed = lib().go_fit_fifo_chain_with_efficiencies(algo, _ptr(avail), _ptr(sched), _ptr(zone), len(avail), _ptr(apps),
                                                           len(apps), _ptr(d), len(d), _ptr(x), len(x), _ptr(res),
                                                           _ptr(off), _ptr(out))
    else:
        failed = lib().go_fit_fifo_chain_ex(algo, int(closed_form), _ptr(avail), _ptr(sched), _ptr(zone), len(avail),
                                            _ptr(apps), len(apps), _ptr(d), len(d), _ptr(x), len(x), _ptr(res), _ptr(off),
                                            _ptr(out))
    return BatchOut(res, off, out[:-1], int(failed), avail)


def avg_packing_efficiency_list(avail, sched, drv, exe, driver_node: int, exec_nodes,
                                reserved_includes_executors: bool = True) -> np.ndarray:
    """ComputeAvgPackingEfficiency over [driver] ++ exec_nodes in slice order -> [CPU, Memory, GPU, Max]."""
    avail = np.ascontiguousarray(avail, dtype=np.int64).reshape(-1, 3)
    sched = np.ascontiguousarray(sched, dtype=np.int64).reshape(-1, 3)
    app = make_apps([drv], [exe], [len(exec_nodes)])
    en = np.ascontiguousarray(exec_nodes, dtype=np.uint32)
    avg = np.zeros(4, dtype=np.float64)
    lib().go_avg_packing_efficiency_list(_ptr(avail), _ptr(sched), len(avail), _ptr(app), driver_node, _ptr(en),
                                         len(en), int(reserved_includes_executors), _ptr(avg))
    return avg


def spark_binpack(algo: int, avail, drv, exe, k: int, driver_order, exec_order, closed_form: bool = False,
                  sched=None, zone=None):
    """One decision. Returns (has_capacity, driver_node, exec_nodes ndarray)."""
    apps = make_apps([drv], [exe], [k])
    out = fit_independent(algo, avail, apps, driver_order, exec_order, closed_form, sched=sched, zone=zone)
    return out.placement(0)


def node_capacity(avail3, reserved3, required3) -> int:
    a = np.asarray(avail3, dtype=np.int64)
    r = np.asarray(reserved3, dtype=np.int64)
    q = np.asarray(required3, dtype=np.int64)
    return int(lib().go_node_capacity(_ptr(a), _ptr(r), _ptr(q)))


def packing_efficiency(avail, sched, drv, exe, driver_node: int, exec_nodes):
    avail = np.ascontiguousarray(avail, dtype=np.int64).reshape(-1, 3)
    sched = np.ascontiguousarray(sched, dtype=np.int64).reshape(-1, 3)
    app = make_apps([drv], [exe], [len(exec_nodes)])
    en = np.ascontiguousarray(exec_nodes, dtype=np.uint32)
    eff = np.zeros((len(avail), 3), dtype=np.float64)
    avg = np.zeros(4, dtype=np.float64)
    lib().go_packing_efficiency(_ptr(avail), _ptr(sched), len(avail), _ptr(app), driver_node, _ptr(en), len(en),
                                _ptr(eff), _ptr(avg))
    return eff, avg


def executor_first_fit(avail, exe, exec_order) -> int:
    avail = np.ascontiguousarray(avail, dtype=np.int64).reshape(-1, 3)
    e = np.asarray(exe, dtype=np.int64)
    x = np.ascontiguousarray(exec_order, dtype=np.uint32)
    return int(lib().go_executor_first_fit(_ptr(avail), len(avail), _ptr(e), _ptr(x), len(x)))


def executor_fit(avail, exe, exec_order, reserved=None, minimal_fragmentation=False, hosts=None) -> int:
    """One executor through rescheduleExecutor's first-fit loop or rescheduleExecutorWithMinimalFragmentation."""
    avail = np.ascontiguousarray(avail, dtype=np.int64).reshape(-1, 3)
    e = np.asarray(exe, dtype=np.int64)
    x = np.ascontiguousarray(exec_order, dtype=np.uint32)
    r = None if reserved is None else np.ascontiguousarray(reserved, dtype=np.int64).reshape(-1, 3)
    if minimal_fragmentation:
        h = None if hosts is None else np.ascontiguousarray(hosts, dtype=np.uint8)
        return int(lib().go_executor_min_frag(_ptr(avail), len(avail), _ptr(r), _ptr(e), _ptr(x), len(x), _ptr(h)))
    return int(lib().go_executor_first_fit_reserved(_ptr(avail), len(avail), _ptr(r), _ptr(e), _ptr(x), len(x)))


@dataclass
class FindNodesOut:
    placed: np.ndarray      # uint32 per request: executors that found a node (<= k)
    exec_off: np.ndarray    # uint64
    exec_nodes: np.ndarray  # uint32, concatenated; request q owns [exec_off[q], exec_off[q] + placed[q])
    adds: np.ndarray        # (n_req, n_nodes) uint32: `reserved[n].Add(exe)` calls = the returned map in units of exe
    avail_after: np.ndarray

    def placement(self, q: int) -> np.ndarray:
        o = int(self.exec_off[q])
        return self.exec_nodes[o:o + int(self.placed[q])]


def find_nodes(avail, exe, k, ordered_nodes, closed_form: bool = False, chained: bool = True) -> FindNodesOut:
    """findNodes (failover.go:412-436) for n_req requests; chained = each request's `reserved` map (over-adds included) is
    subtracted from availableResources before the next one (failover.go:159), else every request sees the same table."""
    avail = np.ascontiguousarray(avail, dtype=np.int64).reshape(-1, 3).copy()
    exe = np.ascontiguousarray(exe, dtype=np.int64).reshape(-1, 3)
    k = np.ascontiguousarray(k, dtype=np.int32).reshape(-1)
    o = np.ascontiguousarray(ordered_nodes, dtype=np.uint32)
    off = exec_offsets(np.maximum(k, 0))
    out = np.zeros(int(np.maximum(k, 0).astype(np.int64).sum()) + 1, dtype=np.uint32)
    placed = np.zeros(len(k), dtype=np.uint32)
    adds = np.zeros((len(k), len(avail)), dtype=np.uint32)
    if chained:
        lib().go_find_nodes_chain(int(closed_form), _ptr(avail), len(avail), _ptr(exe), _ptr(k), len(k), _ptr(o), len(o),
                                  _ptr(placed), _ptr(off), _ptr(out), _ptr(adds))
    else:
        for q in range(len(k)):
            a = avail.copy()
            lib().go_find_nodes_chain(int(closed_form), _ptr(a), len(a), _ptr(exe[q:q + 1]), _ptr(k[q:q + 1]), 1, _ptr(o),
                                      len(o), _ptr(placed[q:q + 1]), _ptr(np.zeros(1, dtype=np.uint64)),
                                      _ptr(out[int(off[q]):]), _ptr(adds[q:q + 1]))
    return FindNodesOut(placed, off, out[:-1], adds, avail)


def fit_maps(algo: int, avail, apps: np.ndarray, driver_order, exec_order, chain: bool, sched=None) -> BatchOut:
    """The FIFO chain (chain=True) or an independent batch on STRING-KEYED MAPS (oracle/gangfit_oracle_maps.cpp): the
    reference's data-structure shape, incl. the per-node efficiency map of every successful pack.  Tightly-pack /
    distribute-evenly.  Same results as fit_fifo_chain / fit_independent; exists for the CPU baseline."""
    avail, d, x = _prep(avail, driver_order, exec_order)
    sched, _ = _aux(avail, sched, None)
    avail = avail.copy()
    apps = np.ascontiguousarray(apps)
    res = np.zeros(len(apps), dtype=RESULT_DTYPE)
    off = exec_offsets(apps["k"])
    out = np.zeros(int(apps["k"].astype(np.int64).sum()) + 1, dtype=np.uint32)
    failed = lib().go_fit_maps(algo, int(chain), _ptr(avail), _ptr(sched), len(avail), _ptr(apps), len(apps), _ptr(d), len(d),
                               _ptr(x), len(x), _ptr(res), _ptr(off), _ptr(out))
    return BatchOut(res, off, out[:-1], int(failed), avail if chain else None)
