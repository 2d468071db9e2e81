set -u
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=16
OUT=gpurun_out/r6bb; mkdir -p $OUT
bash tools/gpu_round.sh r6bb tests bench two
timeout 500 python tools/stress_parity.py 300 67001 > $OUT/stress300.txt 2>&1; echo "stress rc=$?"; tail -2 $OUT/stress300.txt
timeout 300 python tools/stress_sharded.py 120 124001 > $OUT/stress_sharded120.txt 2>&1; echo "sharded rc=$?"; tail -1 $OUT/stress_sharded120.txt
timeout 200 python tools/probe_zoned_batch.py > $OUT/zoned_batch.txt 2>&1; echo "batch rc=$?"
bash tools/gpu_round.sh r6bb prof
