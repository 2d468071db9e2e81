#!/bin/bash
# Timing experiments on the one-wavefront chain: builds of the same sources with pieces of the per-app work removed
# (GF_SOLO_EXP bits: 1 no result store, 2 no run-head store, 4 no commit).  Results are WRONG by construction; only the
# kernel time is of interest.  Build here (no GPU needed), run on the GPU box:
#   tools/micro/solo_experiments.sh build && gpurun -- tools/micro/solo_experiments.sh run
set -u
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
CS=$ROOT/k8s-spark-scheduler_amd/csrc
if [ "${1:-run}" = build ]; then
  for e in ${EXPS:-1 3 7}; do
    mkdir -p $ROOT/tools/micro/exp_$e
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DGF_SOLO_EXP=$e -I $ROOT/include -I $CS $CS/gangfit_kernels.hip \
      $CS/gangfit_snapshot.hip $CS/gangfit_api.cpp -o $ROOT/tools/micro/exp_$e/libgangfit.so 2>/dev/null &
  done
  wait
  ls -la $ROOT/tools/micro/exp_*/libgangfit.so
else
  mkdir -p $ROOT/gpurun_out
  for e in ${EXPS:-1 3 7}; do
    GANGFIT_LIB=$ROOT/tools/micro/exp_$e/libgangfit.so CALLS=5 timeout 100 python $ROOT/tools/fifo_sweep.py > $ROOT/gpurun_out/exp_$e.json 2>/dev/null
    python - <<PY
import json
d=json.load(open("$ROOT/gpurun_out/exp_$e.json"))
print("exp $e", {k:(v["stream_p50_ms"], v["phase_cycles"]) for k,v in d.items() if isinstance(v,dict)})
PY
  done
fi
