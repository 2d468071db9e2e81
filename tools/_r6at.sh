set -u
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=16
OUT=gpurun_out/r6at; mkdir -p $OUT
timeout 1500 python tools/_r6at.py 61001 1300 > $OUT/hunt.txt 2>&1; echo "rc=$?"; tail -30 $OUT/hunt.txt; cat /tmp/cur_seed
