#!/bin/bash
# visit r5au: priority sort with batched key build / 32 count rows per request / zone rank in the kernel, finalize outputs written to pinned memory
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r5au}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"; cd "$ROOT"
export GPU_MAX_HW_QUEUES=16 TMPDIR=/tmp
timeout 600 python -m pytest tests/test_snapshot_build.py tests/test_host_mirror.py -m gpu -q -x --timeout 300 > "$OUT/pytest_snap.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest_snap.log"
( cd k8s-spark-scheduler_amd && timeout 300 ./host_test gpu ) > "$OUT/host_test_gpu.log" 2>&1; echo "host_test rc=$?"; tail -1 "$OUT/host_test_gpu.log"
timeout 200 python tools/probe_snapshot_build.py > "$OUT/snapshot_build.txt" 2>&1; cat "$OUT/snapshot_build.txt"
cd /tmp
for n in 10000 100000; do
  timeout 200 rocprofv3 --kernel-trace --stats -T -f csv -d "$OUT/snap_$n" -o stats -- python $ROOT/tools/probe_snapshot_resident.py $n 200 > "$OUT/snap_$n.log" 2>&1
  grep nodes: "$OUT/snap_$n.log"
  f=$(find "$OUT/snap_$n" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/snap_${n}_kernel_stats.csv" && head -12 "$OUT/snap_${n}_kernel_stats.csv" | cut -c1-110
  rm -rf "$OUT/snap_$n"
done
