#!/bin/bash
# round 4, first GPU visit: validation, the driver's bench line, the headline counter passes, config-5 Filter phases, variants
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r4a
mkdir -p "$OUT"
cd "$ROOT"
export GPU_MAX_HW_QUEUES=16
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; tail -4 "$OUT/pytest_gpu.log"
( cd k8s-spark-scheduler_amd && timeout 300 ./host_test gpu ) > "$OUT/host_test_gpu.log" 2>&1; echo "host_test rc=$?"; tail -2 "$OUT/host_test_gpu.log"
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?"; tail -1 "$OUT/smoke.log"
V=$ROOT/k8s-spark-scheduler_amd/variants
for v in default nopair pairunroll occ96 w8; do
  if [ $v = default ]; then unset GANGFIT_LIB; else export GANGFIT_LIB=$V/libgangfit_$v.so; fi
  timeout 300 python tools/probe_variants.py 2>&1 | tail -1
done | tee "$OUT/variants.txt"
unset GANGFIT_LIB
timeout 300 python tools/probe_c5_filter.py 100 > "$OUT/c5_filter.json" 2> "$OUT/c5_filter.err"; cat "$OUT/c5_filter.json"
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -T -f csv -d "$OUT/c5_trace" -o stats -- python $ROOT/tools/probe_c5_filter.py 30 > "$OUT/c5_trace.log" 2>&1 ); head -30 "$OUT/c5_trace/stats_kernel_stats.csv"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_steps20.json" 2> "$OUT/bench_steps20.err"; echo "bench20 rc=$?"; cut -c1-3000 "$OUT/bench_steps20.json"
bash tools/profile_round.sh r4a headline > "$OUT/profile.log" 2>&1; tail -5 "$OUT/profile.log"
