set -u
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=16
OUT=gpurun_out/r6w; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_zones.py tests/test_gpu_incremental.py -m gpu -q -x --timeout 120 > $OUT/pytest_zones.log 2>&1; echo "pytest zones rc=$?"; tail -4 $OUT/pytest_zones.log
echo "== main" | tee $OUT/probe_zoned.txt
timeout 200 python tools/probe_zoned.py 10000 3 azmajor 2>&1 | grep -v amdgpu.ids | tee -a $OUT/probe_zoned.txt
for v in k8s-spark-scheduler_amd/variants/libgangfit_*.so; do
  [ -f "$v" ] || continue
  echo "== $v" | tee -a $OUT/probe_zoned.txt
  GANGFIT_LIB=$PWD/$v timeout 200 python tools/probe_zoned.py 10000 3 azmajor 2>&1 | grep -v amdgpu.ids | tee -a $OUT/probe_zoned.txt
done
