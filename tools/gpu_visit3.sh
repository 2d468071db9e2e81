#!/bin/bash
# round 4, third GPU visit: the multi-device context by device groups + submitting threads, delta checkpoints (2nd form), reverts
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r4c
mkdir -p "$OUT"
cd "$ROOT"
export GPU_MAX_HW_QUEUES=16
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|FAILED|Error" "$OUT/pytest_gpu.log" | tail -12
( cd k8s-spark-scheduler_amd && timeout 300 ./host_test gpu ) > "$OUT/host_test_gpu.log" 2>&1; echo "host_test rc=$?"; tail -3 "$OUT/host_test_gpu.log"
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?"; tail -1 "$OUT/smoke.log"
timeout 300 python tools/probe_c5_filter.py 100 > "$OUT/c5_filter.json" 2> "$OUT/c5_filter.err"; cat "$OUT/c5_filter.json"
timeout 200 python tools/probe_c5_chain.py 2>&1 | grep tightly | tee "$OUT/c5_chain.txt"
for cfg in headline config4; do
  timeout 120 python tools/group_bench.py --devices 0,0,0,0,0,0,0,0 --config $cfg --steps 20 | tail -1
  GANGFIT_TEST_GROUP_SPLIT=1 timeout 120 python tools/group_bench.py --devices 0,0,0,0,0,0,0,0 --config $cfg --steps 20 | tail -1
done | tee "$OUT/group_bench.txt"
