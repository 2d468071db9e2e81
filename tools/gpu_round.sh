#!/bin/bash
# One GPU visit: the -m gpu parity suite, the torch-free host tests, the driver's bench command and the default one, then
# the profiling recipe.  Everything lands under gpurun_out/<tag>/ (merged back by gpurun).
#   tools/gpu_round.sh <tag> [tests|bench|prof ...]   (default: all three)
set -u
TAG=${1:-r4}
shift || true
WHAT=${*:-tests bench prof}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
export GPU_MAX_HW_QUEUES=16
for w in $WHAT; do
  case $w in
    tests)
      timeout 1500 python -m pytest tests -m gpu -q --timeout 900 --durations=15 > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/pytest_gpu.log"
      grep -E "passed|failed" "$OUT/pytest_gpu.log" | tail -3
      ( cd k8s-spark-scheduler_amd && timeout 300 ./host_test gpu ) > "$OUT/host_test_gpu.log" 2>&1; echo "host_test rc=$?" | tee -a "$OUT/host_test_gpu.log"
      tail -2 "$OUT/host_test_gpu.log"
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?"; tail -1 "$OUT/smoke.log"
      ;;
    bench)
      timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_steps20.json" 2> "$OUT/bench_steps20.err"; echo "bench20 rc=$?"
      cut -c1-700 "$OUT/bench_steps20.json"
      timeout 900 python bench.py --no-extras --no-cpu-baseline > "$OUT/bench_default_headline.json" 2> "$OUT/bench_default.err"; echo "bench rc=$?"
      cut -c1-400 "$OUT/bench_default_headline.json"
      ;;
    prof)
      bash tools/profile_round.sh "$TAG" ${PROF_GROUPS:-} > "$OUT/profile.log" 2>&1; tail -12 "$OUT/profile.log"
      ;;
  esac
done
