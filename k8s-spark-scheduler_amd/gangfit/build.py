"""Build libgangfit.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc.  No JIT cache: the .so sits next to the
package so that it travels with the repository snapshot to the GPU box."""
from __future__ import annotations

import os
import shutil
import subprocess

_PKG_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # k8s-spark-scheduler_amd/
_REPO_ROOT = os.path.dirname(_PKG_ROOT)
CSRC = os.path.join(_PKG_ROOT, "csrc")
LIB_PATH = os.path.join(_PKG_ROOT, "libgangfit.so")
HOST_LIB_PATH = os.path.join(_PKG_ROOT, "libgangfit_host.so")
HOST_BENCH_PATH = os.path.join(_PKG_ROOT, "host_bench")  # end-to-end Filter timing through the host mirror
HOST_TEST_PATH = os.path.join(_PKG_ROOT, "host_test")  # C++ tests of the host mirror (host/tests/host_test.cpp)
INCLUDE = os.path.join(_REPO_ROOT, "include")

_SOURCES = ["gangfit_kernels.hip", "gangfit_snapshot.hip", "gangfit_api.cpp", "gangfit_api_snapshot.cpp", "gangfit_api_fit.cpp",
            "gangfit_api_worker.cpp", "gangfit_api_group.cpp"]
_HEADERS = [os.path.join(CSRC, "gangfit_device.h"), os.path.join(INCLUDE, "gangfit.h")]


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libgangfit cannot be built (ROCm toolchain required)")
    return exe


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_native(force: bool = False, extra_flags=()) -> str:
    """hipcc --offload-arch=gfx950 ... -> k8s-spark-scheduler_amd/libgangfit.so.  The translation units are compiled side by side
    (one hipcc process each: the two kernel files take most of a minute, the five host files a few seconds), then linked."""
    srcs = [os.path.join(CSRC, s) for s in _SOURCES]
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + _HEADERS  # .inc files are #included by the .hip
    if force or extra_flags or _stale(LIB_PATH, deps):
        import tempfile
        from concurrent.futures import ThreadPoolExecutor

        common = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", INCLUDE, "-I", CSRC, *extra_flags]
        with tempfile.TemporaryDirectory(prefix="gangfit_build_") as tmp:
            objs = [os.path.join(tmp, os.path.basename(s) + ".o") for s in srcs]

            def one(pair):
                subprocess.check_call([*common, "-c", pair[0], "-o", pair[1]])

            with ThreadPoolExecutor(max_workers=len(srcs)) as pool:
                list(pool.map(one, zip(srcs, objs)))
            subprocess.check_call([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB_PATH])
    return LIB_PATH


def build_host(force: bool = False) -> str:
    """C++ host mirror of the reference's plug-in interface (string node names, Quantity parsing) on top of the C ABI."""
    host_dir = os.path.join(_PKG_ROOT, "host")
    if not os.path.isdir(host_dir):
        return ""
    srcs = [os.path.join(host_dir, f) for f in sorted(os.listdir(host_dir)) if f.endswith(".cpp")]
    hdrs = [os.path.join(host_dir, f) for f in sorted(os.listdir(host_dir)) if f.endswith(".hpp")]
    if not srcs:
        return ""
    if force or _stale(HOST_LIB_PATH, srcs + hdrs + _HEADERS + [LIB_PATH]):
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-I", INCLUDE, "-I", host_dir, *srcs,
               "-L", _PKG_ROOT, "-lgangfit", "-Wl,-rpath,$ORIGIN", "-o", HOST_LIB_PATH]
        subprocess.check_call(cmd)
    test_src = os.path.join(host_dir, "tests", "host_test.cpp")
    if os.path.exists(test_src) and (force or _stale(HOST_TEST_PATH, [test_src, HOST_LIB_PATH] + hdrs)):
        cmd = ["g++", "-O1", "-std=c++17", "-Wall", "-pthread", "-I", INCLUDE, "-I", host_dir, test_src, "-L", _PKG_ROOT,
               "-lgangfit_host", "-lgangfit", "-Wl,-rpath,$ORIGIN", "-o", HOST_TEST_PATH]
        subprocess.check_call(cmd)
    bench_src = os.path.join(host_dir, "tests", "host_bench.cpp")
    if os.path.exists(bench_src) and (force or _stale(HOST_BENCH_PATH, [bench_src, HOST_LIB_PATH] + hdrs)):
        cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-I", INCLUDE, "-I", host_dir, bench_src, "-L", _PKG_ROOT,
               "-lgangfit_host", "-lgangfit", "-Wl,-rpath,$ORIGIN", "-o", HOST_BENCH_PATH]
        subprocess.check_call(cmd)
    return HOST_LIB_PATH


if __name__ == "__main__":
    print(build_native(force=True))
    print(build_host(force=True))
