set -u
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=16
OUT=gpurun_out/r6as; mkdir -p $OUT
timeout 1300 python tools/stress_parity.py 1200 61001 > $OUT/stress1200.txt 2>&1; echo "stress rc=$?"; tail -1 $OUT/stress1200.txt
timeout 700 python tools/stress_sharded.py 600 71001 > $OUT/stress_sharded600.txt 2>&1; echo "sharded rc=$?"; tail -1 $OUT/stress_sharded600.txt
