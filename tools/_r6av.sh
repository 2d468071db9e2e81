set -u
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=16
OUT=gpurun_out/r6av; mkdir -p $OUT
bash tools/gpu_round.sh r6av tests bench
timeout 700 python tools/stress_parity.py 600 62001 > $OUT/stress600.txt 2>&1; echo "stress rc=$?"; tail -1 $OUT/stress600.txt
timeout 300 python tools/stress_sharded.py 180 111001 > $OUT/stress_sharded180.txt 2>&1; echo "sharded rc=$?"; tail -1 $OUT/stress_sharded180.txt
