#!/bin/bash
# the round's closing visit: whole GPU suite + host tests + smoke, the wide stress, the driver's bench commands, the profiling
# recipe (all groups) and the config-5 Filter probe — everything under gpurun_out/<tag>/ and gpurun_out/prof_<tag>/
set -u
TAG=${1:-r4h}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
export GPU_MAX_HW_QUEUES=16
bash tools/gpu_round.sh $TAG tests 2>&1 | tail -6
GANGFIT_STRESS_SEEDS=400 timeout 900 python -m pytest tests/test_gpu_stress.py -m gpu -q > "$OUT/stress400.log" 2>&1; echo "stress400 rc=$?"; tail -1 "$OUT/stress400.log"
bash tools/gpu_round.sh $TAG bench 2>&1 | grep -E "rc=" 
timeout 300 python tools/probe_c5_filter.py 100 > "$OUT/c5_filter.json" 2> "$OUT/c5_filter.err"; cut -c1-600 "$OUT/c5_filter.json"
timeout 300 python tools/probe_variants.py 2>&1 | tail -1 | tee "$OUT/variants.txt"
bash tools/profile_round.sh $TAG > "$OUT/profile.log" 2>&1; tail -4 "$OUT/profile.log"
