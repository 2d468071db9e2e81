set -u
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=16
OUT=gpurun_out/r6bh; mkdir -p $OUT
bash tools/gpu_round.sh r6bh tests bench
ZONED_SPECS="zb_mf zb_smf" PROF_GROUPS="zoned" bash tools/gpu_round.sh r6bh prof
