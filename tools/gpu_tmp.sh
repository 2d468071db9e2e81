#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
bash tools/gpu_round.sh r5bi tests bench
export GPU_MAX_HW_QUEUES=16
timeout 300 python tools/probe_zoned_batch.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r5bi/zoned_batch.txt; cat gpurun_out/r5bi/zoned_batch.txt
ZONED_SPECS=zb_smf bash tools/profile_round.sh r5bi zoned > gpurun_out/r5bi/profile.log 2>&1; tail -3 gpurun_out/r5bi/profile.log
