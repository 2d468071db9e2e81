set -u
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=16
OUT=gpurun_out/r6af; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_zones.py tests/test_gpu_incremental.py tests/test_host_mirror.py -m gpu -q -x --timeout 600 > $OUT/pytest_zoned.log 2>&1; echo "pytest zoned rc=$?"; tail -3 $OUT/pytest_zoned.log
for i in 1 2; do
  timeout 300 python tools/probe_variants.py chain >> $OUT/variants.txt 2>&1
  GANGFIT_LIB=$PWD/k8s-spark-scheduler_amd/variants/libgangfit_head.so timeout 300 python tools/probe_variants.py chain >> $OUT/variants.txt 2>&1
done
cat $OUT/variants.txt
