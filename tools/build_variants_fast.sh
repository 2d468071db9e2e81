#!/bin/bash
# Several experiment builds of libgangfit that differ only in gangfit_kernels.hip's -D switches: the six other translation
# units are compiled once.   tools/build_variants_fast.sh name1:"-DX=1 -DY=1" name2:"-DZ=1" ...
#   -> k8s-spark-scheduler_amd/variants/libgangfit_<name>.so   (git-ignored; select with GANGFIT_LIB)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOT/k8s-spark-scheduler_amd/variants"
cd "$ROOT/k8s-spark-scheduler_amd/csrc"
TMP=$(mktemp -d /tmp/gangfit_variants_XXXX)
CC="hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I ../../include -I ."
for f in gangfit_snapshot.hip gangfit_api.cpp gangfit_api_snapshot.cpp gangfit_api_fit.cpp gangfit_api_worker.cpp gangfit_api_group.cpp; do
  $CC -c $f -o $TMP/$f.o &
done
wait
n=0
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  ( $CC $flags -c gangfit_kernels.hip -o $TMP/k_$name.o && \
    hipcc --offload-arch=gfx950 -shared -fPIC $TMP/k_$name.o $TMP/gangfit_snapshot.hip.o $TMP/gangfit_api*.o -o "$ROOT/k8s-spark-scheduler_amd/variants/libgangfit_$name.so" && \
    echo "built variants/libgangfit_$name.so ($flags)" ) &
  n=$((n+1)); if [ $((n % 3)) -eq 0 ]; then wait; fi
done
wait
rm -rf $TMP
