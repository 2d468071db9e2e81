#!/bin/bash
# Smoke test of bench.py's N > 1 control flow on a ONE-GPU box: `python bench.py --gpus 2` starts its own two ranks
# (torch.distributed.run on 127.0.0.1); the box has fewer GPUs than ranks, so both ranks share cuda:0 and the process group
# is gloo (RCCL refuses two ranks on one device).  Checks that the run ends with one parseable contract line, n_gpus = 2.
# The numbers are meaningless: the ranks share the device.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
GANGFIT_BENCH_FULL=${GANGFIT_BENCH_FULL:-/tmp/bench_full_two_ranks.json} \
  timeout 500 python bench.py --gpus 2 --steps 20 --warmup 3 --windows 3 --filter-calls 5 --no-cpu-baseline > /tmp/two_ranks.out 2> /tmp/two_ranks.err
rc=$?
echo "rc=$rc"
tail -1 /tmp/two_ranks.out | python -c "import sys, json; l = json.loads(sys.stdin.read()); print('parsed: n_gpus', l['n_gpus'], 'backend', l['config']['backend'], 'rccl_ranks', l['config']['rccl_ranks'], 'value', l['value'])" || tail -5 /tmp/two_ranks.err
