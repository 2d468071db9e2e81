#!/bin/bash
# Smoke test of bench.py's N > 1 control flow on a ONE-GPU box: two ranks, both on cuda:0, gloo instead of RCCL.
# Checks that the run ends and rank 0 prints the one JSON line (numbers are meaningless: the ranks share the device).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29541 WORLD_SIZE=2 GANGFIT_BENCH_BACKEND=gloo
RANK=1 LOCAL_RANK=0 timeout 400 python bench.py --gpus 2 --steps 20 --warmup 3 --windows 3 --filter-calls 5 --no-cpu-baseline > /tmp/rank1.log 2>&1 &
RANK=0 LOCAL_RANK=0 timeout 400 python bench.py --gpus 2 --steps 20 --warmup 3 --windows 3 --filter-calls 5 --no-cpu-baseline
rc=$?
wait
echo "rank0 rc=$rc"; tail -3 /tmp/rank1.log
