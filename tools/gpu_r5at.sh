#!/bin/bash
# visit r5at: per-kernel split of the resident snapshot build (rocprofv3 --kernel-trace --stats), 10 000 and 100 000 nodes
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r5at
mkdir -p "$OUT"
export GPU_MAX_HW_QUEUES=16 TMPDIR=/tmp
cd /tmp
for n in 10000 100000; do
  timeout 200 rocprofv3 --kernel-trace --stats -T -f csv -d "$OUT/snap_$n" -o stats -- python $ROOT/tools/probe_snapshot_resident.py $n 200 > "$OUT/snap_$n.log" 2>&1
  tail -1 "$OUT/snap_$n.log"
  f=$(find "$OUT/snap_$n" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/snap_${n}_kernel_stats.csv" && head -14 "$OUT/snap_${n}_kernel_stats.csv" | cut -c1-150
  rm -rf "$OUT/snap_$n"
done
