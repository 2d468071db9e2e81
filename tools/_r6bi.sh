set -u
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=16
OUT=gpurun_out/r6bi; mkdir -p $OUT
timeout 200 python tools/probe_zoned_batch.py > $OUT/zoned_batch.txt 2>&1; echo "batch rc=$?"; grep "fused=1\|answers" $OUT/zoned_batch.txt
timeout 900 python -m pytest tests/test_gpu_minfrag.py tests/test_gpu_feasible.py tests/test_gpu_zones.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest.log
timeout 300 python tools/stress_parity.py 150 71001 > $OUT/stress150.txt 2>&1; echo "stress rc=$?"; tail -1 $OUT/stress150.txt
