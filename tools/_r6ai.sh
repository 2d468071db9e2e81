set -u
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=16
OUT=gpurun_out/$1; mkdir -p $OUT
for lib in "" head ${2:-}; do
  if [ -n "$lib" ]; then export GANGFIT_LIB=$PWD/k8s-spark-scheduler_amd/variants/libgangfit_$lib.so; else unset GANGFIT_LIB; fi
  echo "== lib ${lib:-default}" >> $OUT/phases.txt
  timeout 300 python tools/probe_minfrag.py azmajor 2>&1 | grep -v amdgpu.ids | head -4 >> $OUT/phases.txt
done
cat $OUT/phases.txt
