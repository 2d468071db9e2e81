#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
bash tools/gpu_round.sh r5bc tests bench
export GPU_MAX_HW_QUEUES=16
timeout 300 python tools/probe_zoned_batch.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r5bc/zoned_batch.txt; cat gpurun_out/r5bc/zoned_batch.txt
ZONED_SPECS=zb_smf bash tools/profile_round.sh r5bc zoned host > gpurun_out/r5bc/profile.log 2>&1; tail -5 gpurun_out/r5bc/profile.log
