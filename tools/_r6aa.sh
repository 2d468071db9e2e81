set -u
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=16
OUT=gpurun_out/r6aa; mkdir -p $OUT
bash tools/gpu_round.sh r6aa tests bench
timeout 400 python tools/stress_parity.py 300 7001 > $OUT/stress300.txt 2>&1; echo "stress rc=$?"; tail -2 $OUT/stress300.txt
