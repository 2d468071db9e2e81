#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r5az; mkdir -p "$OUT"; cd "$ROOT"
GANGFIT_LIB=$ROOT/k8s-spark-scheduler_amd/variants/libgangfit_mfprobe.so timeout 200 python tools/probe_minfrag_phases.py > "$OUT/minfrag_phases.txt" 2>&1; cat "$OUT/minfrag_phases.txt"
