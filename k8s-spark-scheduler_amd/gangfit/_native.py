"""ctypes binding of libgangfit.so — the C ABI declared in include/gangfit.h.

There is NO fallback: if the shared library is missing or a call fails this module raises.  The HIP path is the only
product path.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
from typing import Optional

import numpy as np

from . import build as _build

GF_OK = 0
GF_ERR_NO_DEVICE, GF_ERR_HIP, GF_ERR_INVALID, GF_ERR_CAPACITY, GF_ERR_STATE, GF_ERR_UNSUPPORTED = -1, -2, -3, -4, -5, -6
GF_NO_NODE = 0xFFFFFFFF
GF_MAX_K = 1 << 20
GF_ALGO_TIGHTLY_PACK = 0
GF_ALGO_DISTRIBUTE_EVENLY = 1
GF_ALGO_MINIMAL_FRAGMENTATION = 2
GF_ALGO_AZ_AWARE_TIGHTLY_PACK = 3
GF_ALGO_SINGLE_AZ_TIGHTLY_PACK = 4
GF_ALGO_SINGLE_AZ_MINIMAL_FRAGMENTATION = 5
GF_MODE_INDEPENDENT = 0
GF_MODE_FIFO_CHAIN = 1
GF_APP_SKIPPABLE = 1

APP_DTYPE = np.dtype(
    [("drv", "<i8", (3,)), ("exe", "<i8", (3,)), ("k", "<i4"), ("flags", "<u4"), ("exec_off", "<u8")], align=True
)
RESULT_DTYPE = np.dtype(
    [("has_capacity", "<i4"), ("driver_node", "<u4"), ("exec_len", "<u4"), ("evaluated", "<u4")], align=True
)
assert APP_DTYPE.itemsize == 64 and RESULT_DTYPE.itemsize == 16

# every symbol include/gangfit.h declares (tests check that the library exports all of them)
GF_RESIDENT_USAGE = 0xFFFFFFFF
GF_ANY_ZONE = 0xFFFFFFFF
GF_WORKER_LEAVE_AFTER = 2

EXPORTED_SYMBOLS = [
    "gf_version", "gf_init", "gf_destroy", "gf_last_error", "gf_snapshot_set", "gf_orders_set", "gf_fit_batch", "gf_fit_feasible",
    "gf_fit_batch_dev", "gf_spark_binpack", "gf_residual_get", "gf_timer_begin", "gf_timer_end", "gf_scan_stats", "gf_chain_profile",
    "gf_selftest", "gf_device_info_get", "gf_zones_set", "gf_avg_packing_efficiency", "gf_packing_efficiencies",
    "gf_hbm_probe", "gf_executor_fit", "gf_executor_fit_zoned", "gf_snapshot_build", "gf_snapshot_get", "gf_shard_set", "gf_shard_partials_dev", "gf_shard_drivers_dev", "gf_shard_emit_dev", "gf_shard_finish_dev",
    "gf_find_nodes", "gf_ctx_lock", "gf_ctx_unlock", "gf_launch_floor",
    "gf_graph_begin", "gf_graph_end", "gf_graph_launch", "gf_graph_destroy", "gf_cluster_set", "gf_snapshot_build_resident",
    "gf_usage_reset", "gf_usage_apply", "gf_set_option", "gf_chain_cache_stats", "gf_generation", "gf_shard_count", "gf_ctx_view",
    "gf_worker_fit", "gf_worker_submit_dev", "gf_worker_wait", "gf_worker_stop", "gf_worker_stats", "gf_worker_geometry", "gf_worker_kernel_time", "gf_call_phases",
]


class DeviceInfo(C.Structure):
    _fields_ = [("name", C.c_char * 128), ("arch", C.c_char * 64), ("compute_units", C.c_int32),
                ("lds_bytes_per_cu", C.c_int32), ("wavefront_size", C.c_int32), ("clock_khz", C.c_int32),
                ("hbm_bytes", C.c_int64)]


class WorkerBatch(C.Structure):
    """gf_worker_batch (include/gangfit.h)."""
    _fields_ = [("n_apps", C.c_uint32), ("flags", C.c_uint32), ("d_apps", C.c_void_p), ("d_results", C.c_void_p),
                ("d_exec_nodes", C.c_void_p), ("exec_nodes_len", C.c_uint64)]


GF_WORKER_HOST_OUTPUTS = 1


class GangfitError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"libgangfit error {code}: {message}")
        self.code = code


_lib: Optional[C.CDLL] = None

# GPU_MAX_HW_QUEUES=16 (sixteen hardware queues, so that concurrent views' chains do not share one) is the DEPLOYMENT's
# setting ("Deployment"): neither the library nor this package changes the process environment.  The HIP
# runtime reads the variable when it initialises, so whoever starts the process sets it — tests/conftest.py and bench.py do
# so explicitly, host_test.cpp's main() does, a Go host's unit file does.  gf_ctx_view leaves a note in gf_last_error when it
# finds fewer than 8.


def library_path() -> str:
    """The in-tree library; GANGFIT_LIB points tuning experiments at another build of the same sources."""
    return os.environ.get("GANGFIT_LIB") or _build.LIB_PATH


def load() -> C.CDLL:
    """dlopen the in-tree libgangfit.so; raises if it has not been built (see __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    # One HIP/HSA runtime per process: PyTorch-ROCm bundles its own libamdhip64/libhsa-runtime64, and a process that
    # loads /opt/rocm's copy first and torch's second ends up with two HSA runtimes (torch then sees no GPU).  When torch
    # is installed, load it first so that libgangfit.so binds to the runtime torch already brought in.  A Go host has no
    # torch and simply uses /opt/rocm's runtime.
    if "torch" not in sys.modules and os.environ.get("GANGFIT_NO_TORCH") != "1":
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    path = library_path()
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: run `python __graft_entry__.py` (build()) first — there is no CPU fallback")
    L = C.CDLL(path)
    p, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int
    L.gf_version.restype = i32
    L.gf_init.restype = i32
    L.gf_init.argtypes = [p, i32, C.POINTER(p)]
    L.gf_destroy.restype = None
    L.gf_destroy.argtypes = [p]
    L.gf_last_error.restype = C.c_char_p
    L.gf_last_error.argtypes = [p]
    L.gf_snapshot_set.restype = i32
    L.gf_snapshot_set.argtypes = [p, u32, p, p, p, p, p, p]
    L.gf_orders_set.restype = i32
    L.gf_orders_set.argtypes = [p, p, u32, p, u32]
    L.gf_fit_batch.restype = i32
    L.gf_fit_batch.argtypes = [p, i32, i32, u32, p, p, p, u64, p]
    L.gf_fit_batch_dev.restype = i32
    L.gf_fit_batch_dev.argtypes = [p, i32, i32, u32, p, p, p, u64, p, p]
    L.gf_spark_binpack.restype = i32
    L.gf_spark_binpack.argtypes = [p, i32, p, p, p, u64]
    L.gf_zones_set.restype = i32
    L.gf_zones_set.argtypes = [p, p]
    L.gf_avg_packing_efficiency.restype = i32
    L.gf_avg_packing_efficiency.argtypes = [p, i32, u32, p, p, p, u64, p]
    L.gf_packing_efficiencies.restype = i32
    L.gf_packing_efficiencies.argtypes = [p, i32, p, p, p, p]
    L.gf_hbm_probe.restype = i32
    L.gf_hbm_probe.argtypes = [p, u64, u32, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.gf_graph_begin.restype = i32
    L.gf_graph_begin.argtypes = [p, p]
    L.gf_graph_end.restype = i32
    L.gf_graph_end.argtypes = [p, p, C.POINTER(p)]
    L.gf_graph_launch.restype = i32
    L.gf_graph_launch.argtypes = [p, p, p]
    L.gf_graph_destroy.restype = None
    L.gf_graph_destroy.argtypes = [p, p]
    L.gf_launch_floor.restype = i32
    L.gf_launch_floor.argtypes = [p, p, u32, C.POINTER(C.c_float)]
    L.gf_snapshot_build.restype = i32
    L.gf_snapshot_build.argtypes = [p, u32, p, p, p, p, p, p, u32, p, p, p, p, p, p, u32, p, p, p, p, p, p, p]
    L.gf_cluster_set.restype = i32
    L.gf_cluster_set.argtypes = [p, u32, p, p, p, p, p, p, p, p, u32, p]
    L.gf_usage_reset.restype = i32
    L.gf_usage_reset.argtypes = [p]
    L.gf_usage_apply.restype = i32
    L.gf_usage_apply.argtypes = [p, u32, p, p, p, p, i32]
    L.gf_snapshot_build_resident.restype = i32
    L.gf_snapshot_build_resident.argtypes = [p, u32, p, p, p, p, p, p, p, p, p, p, p]
    L.gf_snapshot_get.restype = i32
    L.gf_snapshot_get.argtypes = [p, p, p]
    L.gf_executor_fit.restype = i32
    L.gf_executor_fit.argtypes = [p, i32, u32, p, p, p, p]
    L.gf_executor_fit_zoned.restype = i32
    L.gf_executor_fit_zoned.argtypes = [p, i32, u32, p, p, p, p, p, p]
    L.gf_ctx_lock.restype = None
    L.gf_ctx_lock.argtypes = [p]
    L.gf_ctx_unlock.restype = None
    L.gf_ctx_unlock.argtypes = [p]
    L.gf_find_nodes.restype = i32
    L.gf_find_nodes.argtypes = [p, i32, u32, p, p, p, p, u64, p]
    L.gf_shard_set.restype = i32
    L.gf_shard_set.argtypes = [p, u32, u32]
    L.gf_shard_partials_dev.restype = i32
    L.gf_shard_partials_dev.argtypes = [p, i32, u32, p, p, p]
    L.gf_shard_drivers_dev.restype = i32
    L.gf_shard_drivers_dev.argtypes = [p, i32, u32, p, p, p, p]
    L.gf_shard_emit_dev.restype = i32
    L.gf_shard_emit_dev.argtypes = [p, i32, u32, p, p, p, p, p, u64, p]
    L.gf_shard_finish_dev.restype = i32
    L.gf_shard_finish_dev.argtypes = [p, i32, u32, p, p, p, p, p, u64, p]
    L.gf_residual_get.restype = i32
    L.gf_residual_get.argtypes = [p, p]
    L.gf_timer_begin.restype = i32
    L.gf_timer_begin.argtypes = [p, p]
    L.gf_timer_end.restype = i32
    L.gf_timer_end.argtypes = [p, C.POINTER(C.c_float)]
    L.gf_scan_stats.restype = i32
    L.gf_scan_stats.argtypes = [p, i32, i32, p]
    L.gf_selftest.restype = i32
    L.gf_selftest.argtypes = [p, u64, u32, C.POINTER(u32)]
    L.gf_device_info_get.restype = i32
    L.gf_device_info_get.argtypes = [p, C.POINTER(DeviceInfo)]
    L.gf_set_option.restype = i32
    L.gf_set_option.argtypes = [p, C.c_char_p, C.c_int64]
    L.gf_ctx_view.restype = i32
    L.gf_ctx_view.argtypes = [p, C.POINTER(p)]
    L.gf_worker_fit.restype = i32
    L.gf_worker_fit.argtypes = [p, i32, u32, p, p, p, u64]
    L.gf_worker_submit_dev.restype = i32
    L.gf_worker_submit_dev.argtypes = [p, i32, u32, p, p]
    L.gf_worker_wait.restype = i32
    L.gf_worker_wait.argtypes = [p, u64, u32]
    L.gf_worker_stop.restype = i32
    L.gf_worker_stop.argtypes = [p]
    L.gf_worker_stats.restype = i32
    L.gf_worker_stats.argtypes = [p, p]
    L.gf_worker_geometry.restype = i32
    L.gf_worker_geometry.argtypes = [p, p]
    L.gf_fit_feasible.restype = i32
    L.gf_fit_feasible.argtypes = [p, i32, u32, p, p]
    L.gf_chain_profile.restype = i32
    L.gf_chain_profile.argtypes = [p, p]
    L.gf_call_phases.restype = i32
    L.gf_call_phases.argtypes = [p, C.POINTER(C.c_double)]
    L.gf_worker_kernel_time.restype = i32
    L.gf_worker_kernel_time.argtypes = [p, C.POINTER(C.c_float), C.POINTER(C.c_uint64)]
    L.gf_shard_count.restype = i32
    L.gf_shard_count.argtypes = [p]
    L.gf_generation.restype = i32
    L.gf_generation.argtypes = [p, p]
    L.gf_chain_cache_stats.restype = i32
    L.gf_chain_cache_stats.argtypes = [p, i32, p]
    _lib = L
    return L


def ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)
