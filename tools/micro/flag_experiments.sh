#!/bin/bash
# Compiler-flag experiments on the chain kernels: the same sources built with extra flags (one library per variant under
# tools/micro/exp_<name>/, git-ignored), timed on the GPU box through GANGFIT_LIB.
#   tools/micro/flag_experiments.sh build   (here, no GPU)      tools/micro/flag_experiments.sh run   (on the GPU box)
set -u
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
CS=$ROOT/k8s-spark-scheduler_amd/csrc
declare -A FLAGS=(
  [ifcvt]="-mllvm -amdgpu-early-ifcvt=1"
  [skip64]="-mllvm -amdgpu-skip-threshold=64"
  [ilp]="-mllvm -amdgpu-sched-strategy=max-ilp"
  [o2]="-O2"
  [nopostsched]="-mllvm -enable-post-misched=0"
  [ifcvt_skip]="-mllvm -amdgpu-early-ifcvt=1 -mllvm -amdgpu-skip-threshold=64"
)
if [ "${1:-run}" = build ]; then
  for name in "${!FLAGS[@]}"; do
    mkdir -p $ROOT/tools/micro/exp_$name
    ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared ${FLAGS[$name]} -I $ROOT/include -I $CS $CS/gangfit_kernels.hip \
        $CS/gangfit_snapshot.hip $CS/gangfit_api.cpp -o $ROOT/tools/micro/exp_$name/libgangfit.so > $ROOT/tools/micro/exp_$name/build.log 2>&1 \
        && echo "built $name" || echo "FAILED $name: $(tail -2 $ROOT/tools/micro/exp_$name/build.log | head -1)" ) &
  done
  wait
else
  for name in base "${!FLAGS[@]}"; do
    lib=$ROOT/tools/micro/exp_$name/libgangfit.so
    [ "$name" = base ] && lib=$ROOT/k8s-spark-scheduler_amd/libgangfit.so
    [ -f "$lib" ] || continue
    echo "== $name"
    GANGFIT_LIB=$lib timeout 120 python $ROOT/tools/micro/probe_chain_times.py 2>&1 | grep -v amdgpu.ids
  done
fi
