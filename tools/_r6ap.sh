set -u
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=16
OUT=gpurun_out/r6ap; mkdir -p $OUT
timeout 400 python tools/stress_sharded.py 240 31001 > $OUT/stress_sharded.txt 2>&1; echo "rc=$?"; tail -3 $OUT/stress_sharded.txt
