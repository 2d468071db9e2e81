"""gangfit — MI355X-native gang-scheduling bin-packer for the k8s-spark-scheduler extender.

Python side of the package: ctypes marshalling over the C ABI (include/gangfit.h) for tests and bench.py, plus the
synthetic workload generators.  The product is libgangfit.so (csrc/: hand-written gfx950 HIP kernels + C ABI) and
libgangfit_host.so (host/: C++ mirror of the reference's plug-in interface).
"""
from . import _native, build, workloads  # noqa: F401
from ._native import (  # noqa: F401
    GF_ALGO_AZ_AWARE_TIGHTLY_PACK,
    GF_ALGO_DISTRIBUTE_EVENLY,
    GF_ALGO_MINIMAL_FRAGMENTATION,
    GF_ALGO_SINGLE_AZ_MINIMAL_FRAGMENTATION,
    GF_ALGO_SINGLE_AZ_TIGHTLY_PACK,
    GF_ALGO_TIGHTLY_PACK,
    GF_APP_SKIPPABLE,
    GF_MODE_FIFO_CHAIN,
    GF_MODE_INDEPENDENT,
    GF_NO_NODE,
    GangfitError,
)
from .context import BatchOut, Context, make_apps, with_offsets  # noqa: F401
