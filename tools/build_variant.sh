#!/bin/bash
# Experiment builds of libgangfit (same sources, extra -D switches): tools/build_variant.sh <name> <flags...>
#   -> k8s-spark-scheduler_amd/variants/libgangfit_<name>.so   (git-ignored; travels to the GPU box; select with GANGFIT_LIB)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
mkdir -p "$ROOT/k8s-spark-scheduler_amd/variants"
cd "$ROOT/k8s-spark-scheduler_amd/csrc"
TMP=$(mktemp -d /tmp/gangfit_variant_XXXX)
for f in gangfit_kernels.hip gangfit_snapshot.hip gangfit_api.cpp gangfit_api_snapshot.cpp gangfit_api_fit.cpp gangfit_api_worker.cpp gangfit_api_group.cpp; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I ../../include -I . "$@" -c $f -o $TMP/$f.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC $TMP/*.o -o "$ROOT/k8s-spark-scheduler_amd/variants/libgangfit_$name.so"
rm -rf $TMP
echo "built variants/libgangfit_$name.so ($*)"
