set -u
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=16
OUT=gpurun_out/$1; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_minfrag.py tests/test_gpu_incremental.py tests/test_gpu_zones.py tests/test_gpu_fullsize.py -m gpu -q -x --timeout 600 > $OUT/pytest_mf.log 2>&1; echo "pytest mf rc=$?"; tail -5 $OUT/pytest_mf.log
for i in 1 2; do
  timeout 300 python tools/probe_variants.py chain >> $OUT/variants.txt 2>&1
  GANGFIT_LIB=$PWD/k8s-spark-scheduler_amd/variants/libgangfit_head.so timeout 300 python tools/probe_variants.py chain >> $OUT/variants.txt 2>&1
done
grep -v amdgpu.ids $OUT/variants.txt
