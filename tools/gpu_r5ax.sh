#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r5ax; mkdir -p "$OUT"; cd "$ROOT"
{ echo "# new"; timeout 200 python tools/probe_minfrag_batch.py; echo "# head"; GANGFIT_LIB=$ROOT/k8s-spark-scheduler_amd/variants/libgangfit_head.so timeout 200 python tools/probe_minfrag_batch.py; } > "$OUT/minfrag_batch.txt" 2>&1; cat "$OUT/minfrag_batch.txt"
