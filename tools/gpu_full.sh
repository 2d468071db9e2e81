#!/bin/bash
# the whole GPU suite + the torch-free host tests + the variant probe (independent batches and cold chains)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/full_${1:-x}
mkdir -p "$OUT"
cd "$ROOT"
export GPU_MAX_HW_QUEUES=16
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|FAILED|Error" "$OUT/pytest_gpu.log" | tail -8
( cd k8s-spark-scheduler_amd && timeout 300 ./host_test gpu ) > "$OUT/host_test_gpu.log" 2>&1; echo "host_test rc=$?"; tail -1 "$OUT/host_test_gpu.log"
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?"; tail -1 "$OUT/smoke.log"
timeout 300 python tools/probe_variants.py 2>&1 | tail -1 | tee "$OUT/variants.txt"
