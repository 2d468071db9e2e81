set -u
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=16
OUT=gpurun_out/r6aq; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_group.py -m gpu -q -x --timeout 600 > $OUT/pytest_shard.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $OUT/pytest_shard.log | tail -2
( cd k8s-spark-scheduler_amd && timeout 300 ./host_test gpu ) > $OUT/host_test_gpu.log 2>&1; echo "host_test rc=$?"; tail -1 $OUT/host_test_gpu.log
timeout 200 python tools/stress_sharded.py 60 51001 > $OUT/stress_sharded.txt 2>&1; echo "rc=$?"; tail -1 $OUT/stress_sharded.txt
GANGFIT_BENCH_FULL="$OUT/bench_full_two_ranks.json" bash tools/smoke_two_ranks_one_gpu.sh > $OUT/two_ranks.log 2>&1; tail -2 $OUT/two_ranks.log
