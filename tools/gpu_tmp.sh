#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r5bh; mkdir -p "$OUT"; cd "$ROOT"
export GPU_MAX_HW_QUEUES=16
timeout 600 python -m pytest tests/test_gpu_minfrag.py -m gpu -q -x --timeout 600 2>&1 | tail -2
timeout 300 python tools/probe_zoned_batch.py 2>&1 | grep -v amdgpu.ids | grep "minimal" > "$OUT/zoned_batch.txt"; cat "$OUT/zoned_batch.txt"
timeout 200 python tools/probe_minfrag_batch.py 2>&1 | grep -v amdgpu.ids > "$OUT/minfrag_batch.txt"; cat "$OUT/minfrag_batch.txt"
