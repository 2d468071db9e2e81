set -u
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=16
OUT=gpurun_out/r6bc; mkdir -p $OUT
PROBE_ORDER=scattered timeout 300 python tools/probe_zoned_parts.py single-az-tightly-pack > $OUT/zoned_parts_scattered.txt 2>&1; echo "rc=$?"; cat $OUT/zoned_parts_scattered.txt
timeout 300 python tools/probe_zoned_parts.py single-az-tightly-pack minimal-fragmentation > $OUT/zoned_parts.txt 2>&1; echo "rc=$?"; cat $OUT/zoned_parts.txt
