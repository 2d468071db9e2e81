set -u
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=16
OUT=gpurun_out/r6ba; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_zones.py tests/test_gpu_feasible.py tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -q -x > $OUT/pytest_zones.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_zones.log
timeout 300 python tools/probe_zoned_parts.py > $OUT/zoned_parts.txt 2>&1; echo "parts rc=$?"; grep -v "^tightly" $OUT/zoned_parts.txt
timeout 200 python tools/probe_feasible.py > $OUT/feasible.txt 2>&1; echo "feasible rc=$?"; tail -12 $OUT/feasible.txt
timeout 500 python tools/stress_parity.py 240 66001 > $OUT/stress240.txt 2>&1; echo "stress rc=$?"; tail -2 $OUT/stress240.txt
