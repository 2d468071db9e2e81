set -u
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=16
OUT=gpurun_out/r6x; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_minfrag.py tests/test_gpu_feasible.py tests/test_gpu_parity.py tests/test_gpu_zones.py -m gpu -q -x --timeout 300 > $OUT/pytest_mf.log 2>&1; echo "pytest mf rc=$?"; tail -2 $OUT/pytest_mf.log
timeout 300 python tools/probe_minfrag_batch.py 2>&1 | grep -v amdgpu.ids | grep "K <=      8\|100000" | tee $OUT/minfrag_batch.txt
timeout 300 python tools/probe_zoned_batch.py 2>&1 | grep -v amdgpu.ids | grep -i "minimal" | tee -a $OUT/minfrag_batch.txt
for v in k8s-spark-scheduler_amd/variants/libgangfit_*.so; do
  [ -f "$v" ] || continue
  echo "== $v" | tee -a $OUT/minfrag_batch.txt
  GANGFIT_LIB=$PWD/$v timeout 300 python tools/probe_zoned_batch.py 2>&1 | grep -v amdgpu.ids | grep "^minimal" | tee -a $OUT/minfrag_batch.txt
done
