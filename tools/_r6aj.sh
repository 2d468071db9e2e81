set -u
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=16
OUT=gpurun_out/$1; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_minfrag.py tests/test_gpu_incremental.py tests/test_gpu_zones.py tests/test_gpu_fullsize.py -m gpu -q -x --timeout 600 > $OUT/pytest_mf.log 2>&1; echo "pytest mf rc=$?"; tail -3 $OUT/pytest_mf.log
for lib in "" head; do
  if [ -n "$lib" ]; then export GANGFIT_LIB=$PWD/k8s-spark-scheduler_amd/variants/libgangfit_$lib.so; else unset GANGFIT_LIB; fi
  echo "== lib ${lib:-default}" >> $OUT/phases.txt
  timeout 300 python tools/probe_minfrag.py azmajor 2>&1 | grep -v amdgpu.ids | head -4 >> $OUT/phases.txt
done
cat $OUT/phases.txt
