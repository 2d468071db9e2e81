set -u
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=16
OUT=gpurun_out/r6ax; mkdir -p $OUT
timeout 300 python tools/probe_zoned_parts.py > $OUT/zoned_parts.txt 2>&1; echo "parts rc=$?"; cat $OUT/zoned_parts.txt
timeout 400 python tools/stress_sharded.py 240 112001 > $OUT/stress_sharded240.txt 2>&1; echo "sharded rc=$?"; tail -1 $OUT/stress_sharded240.txt
