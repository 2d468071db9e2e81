set -u
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=16
OUT=gpurun_out/$1; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_sharded.py -m gpu -q -x --timeout 600 -k "compact_view or headline_size" > $OUT/pytest_shard2.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|Error|assert" $OUT/pytest_shard2.log | tail -8
