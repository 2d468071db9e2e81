set -u
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=16
OUT=gpurun_out/r6ay; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_zones.py tests/test_gpu_feasible.py tests/test_gpu_fullsize.py -m gpu -q -x > $OUT/pytest_zones.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_zones.log
timeout 300 python tools/probe_zoned_parts.py > $OUT/zoned_parts.txt 2>&1; echo "parts rc=$?"; grep "single-az" $OUT/zoned_parts.txt
timeout 200 python tools/probe_zoned_batch.py > $OUT/zoned_batch.txt 2>&1; echo "batch rc=$?"; cat $OUT/zoned_batch.txt
timeout 500 python tools/stress_parity.py 240 65001 > $OUT/stress240.txt 2>&1; echo "stress rc=$?"; tail -2 $OUT/stress240.txt
