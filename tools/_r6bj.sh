set -u
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=16
bash tools/gpu_round.sh r6bj tests bench
