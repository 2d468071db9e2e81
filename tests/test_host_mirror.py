"""The C++ host mirror (k8s-spark-scheduler_amd/host/: Quantity parsing, sparkResources, FIFO predecessor set,
NodeSorter.PotentialNodes, snapshot construction, reservations, the SparkBinPackFunction seam and selectDriverNode) is
tested by a C++ program written after the reference's own Go tests (host/tests/host_test.cpp).  This file runs it:
`cpu` needs no GPU, `gpu` drives the device through the C ABI the way the Go shim would (no torch in that process)."""
import os
import subprocess

import pytest

from gangfit import build


def _binary():
    build.build_native()
    build.build_host()
    assert os.path.exists(build.HOST_TEST_PATH), "host_test was not built"
    return build.HOST_TEST_PATH


def _run(mode):
    p = subprocess.run([_binary(), mode], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert p.returncode == 0 and " 0 failed" in p.stdout, p.stdout[-4000:]
    return p.stdout


def test_host_mirror_cpu_half():
    out = _run("cpu")
    assert "cpu:" in out


@pytest.mark.gpu
def test_host_mirror_through_the_device():
    out = _run("gpu")
    assert "gpu:" in out


@pytest.mark.gpu
@pytest.mark.parametrize("packer", ["tightly-pack", "distribute-evenly", "single-az-tightly-pack", "az-aware-tightly-pack",
                                    "single-az-minimal-fragmentation"])
def test_flat_route_agrees_with_map_route(packer):
    """selectDriverNode on string-keyed maps (the reference's shape) and selectDriverNodeFlat (snapshot built on the
    device by gf_snapshot_build) must produce the same Filter result and the same reservation; host_bench exits 1 if
    they do not."""
    _binary()
    assert os.path.exists(build.HOST_BENCH_PATH)
    p = subprocess.run([build.HOST_BENCH_PATH, "700", "120", "90", packer], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=600)
    assert p.returncode == 0 and "routes agree: yes" in p.stdout, p.stdout[-3000:]
