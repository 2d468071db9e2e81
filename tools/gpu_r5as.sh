#!/bin/bash
# visit r5as: snapshot build with one clearing launch / per-workgroup range atomics / candidate-round gcd, worker occupancy query cached
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r5as
mkdir -p "$OUT"; cd "$ROOT"
export GPU_MAX_HW_QUEUES=16
timeout 600 python -m pytest tests/test_snapshot_build.py tests/test_host_mirror.py -m gpu -q -x --timeout 300 > "$OUT/pytest_snap.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest_snap.log"
( cd k8s-spark-scheduler_amd && timeout 300 ./host_test gpu ) > "$OUT/host_test_gpu.log" 2>&1; echo "host_test rc=$?"; tail -2 "$OUT/host_test_gpu.log"
{ echo "# new"; timeout 200 python tools/probe_snapshot_build.py; echo "# head"; GANGFIT_LIB=$ROOT/k8s-spark-scheduler_amd/variants/libgangfit_head.so timeout 200 python tools/probe_snapshot_build.py; } > "$OUT/snapshot_build.txt" 2>&1; cat "$OUT/snapshot_build.txt"
{ echo "# new"; timeout 200 python tools/probe_worker_sets.py 11:21 2>&1 | grep sets; echo "# head"; GANGFIT_LIB=$ROOT/k8s-spark-scheduler_amd/variants/libgangfit_head.so timeout 200 python tools/probe_worker_sets.py 11:21 2>&1 | grep sets; } > "$OUT/worker_window.txt" 2>&1; cat "$OUT/worker_window.txt"
