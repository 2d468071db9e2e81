set -u
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=16
OUT=gpurun_out/$1; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_group.py tests/test_gpu_fullsize.py -m gpu -q -x --timeout 600 > $OUT/pytest_shard.log 2>&1; echo "pytest shard rc=$?"; tail -4 $OUT/pytest_shard.log
( cd k8s-spark-scheduler_amd && timeout 300 ./host_test gpu ) > $OUT/host_test_gpu.log 2>&1; echo "host_test rc=$?"; tail -2 $OUT/host_test_gpu.log
GANGFIT_BENCH_FULL="$OUT/bench_full_steps20.json" timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_steps20.json 2> $OUT/bench_steps20.err; echo "bench rc=$?"
python - <<'PY'
import json,sys,os
d=json.loads(open(os.path.join("gpurun_out",os.environ.get("TAG","r6al"),"bench_steps20.json")).read().strip().split("\n")[-1])
print(d["value"], d["ms_per_step"], json.dumps(d["config"]["node_sharded"]))
PY
