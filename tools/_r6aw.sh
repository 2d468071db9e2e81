set -u
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=16
OUT=gpurun_out/r6aw; mkdir -p $OUT
timeout 300 python tools/probe_worker_sets.py 11:21 10:23 16:14 20:11 20:12 13:18 > $OUT/worker_sets.txt 2>&1; echo "sets rc=$?"; cat $OUT/worker_sets.txt
timeout 1000 python tools/stress_parity.py 600 63001 > $OUT/stress600.txt 2>&1; echo "stress rc=$?"; tail -2 $OUT/stress600.txt
timeout 400 python tools/stress_sharded.py 240 112001 > $OUT/stress_sharded240.txt 2>&1; echo "sharded rc=$?"; tail -1 $OUT/stress_sharded240.txt
