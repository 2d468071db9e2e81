#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r5bb; mkdir -p "$OUT"; cd "$ROOT"
{ echo "# shipped (4 chunks requested together)"; timeout 200 python tools/probe_minfrag_batch.py; echo "# -DGF_MF_AHEAD=8"; GANGFIT_LIB=$ROOT/k8s-spark-scheduler_amd/variants/libgangfit_mfahead8.so timeout 200 python tools/probe_minfrag_batch.py; } 2>&1 | grep -v amdgpu.ids > "$OUT/minfrag_ahead.txt"; cat "$OUT/minfrag_ahead.txt"
