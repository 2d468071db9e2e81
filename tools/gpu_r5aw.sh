#!/bin/bash
# visit r5aw: minimal-fragmentation independent batch with the capacity histogram (one pass for the common ending)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r5aw}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"; cd "$ROOT"
export GPU_MAX_HW_QUEUES=16 TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_minfrag.py tests/test_gpu_zones.py tests/test_gpu_feasible.py tests/test_executor_fit.py tests/test_gpu_stress.py -m gpu -q -x --timeout 600 > "$OUT/pytest_mf.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest_mf.log"
{ echo "# new"; timeout 300 python tools/probe_zoned_batch.py; echo "# head"; GANGFIT_LIB=$ROOT/k8s-spark-scheduler_amd/variants/libgangfit_head.so timeout 300 python tools/probe_zoned_batch.py; } > "$OUT/zoned_batch.txt" 2>&1; cat "$OUT/zoned_batch.txt"
