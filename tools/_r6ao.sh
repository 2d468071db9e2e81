set -u
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=16
OUT=gpurun_out/r6ao; mkdir -p $OUT
bash tools/gpu_round.sh r6ao tests bench two
timeout 400 python tools/stress_parity.py 300 23001 > $OUT/stress300.txt 2>&1; echo "stress rc=$?"; tail -2 $OUT/stress300.txt
bash tools/gpu_round.sh r6ao prof
