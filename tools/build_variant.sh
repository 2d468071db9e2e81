#!/bin/bash
# Experiment builds of libgangfit (same sources, extra -D switches): tools/build_variant.sh <name> <flags...>
#   -> k8s-spark-scheduler_amd/variants/libgangfit_<name>.so   (git-ignored; travels to the GPU box; select with GANGFIT_LIB)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
mkdir -p "$ROOT/k8s-spark-scheduler_amd/variants"
cd "$ROOT/k8s-spark-scheduler_amd/csrc"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I ../../include -I . "$@" gangfit_kernels.hip gangfit_snapshot.hip gangfit_api.cpp \
  -o "$ROOT/k8s-spark-scheduler_amd/variants/libgangfit_$name.so"
echo "built variants/libgangfit_$name.so ($*)"
