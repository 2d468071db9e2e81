#!/bin/bash
# chain-kernel iteration: parity of everything that touches the chains, then the cold-chain times and the solo kernel's phases
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/chain_${1:-x}
mkdir -p "$OUT"
cd "$ROOT"
export GPU_MAX_HW_QUEUES=16
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_incremental.py tests/test_gpu_stress.py tests/test_gpu_units.py tests/test_gpu_fullsize.py tests/test_golden.py tests/test_gpu_zones.py tests/test_gpu_minfrag.py -m gpu -q -x --timeout 600 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|FAILED|Error" "$OUT/pytest.log" | tail -5
for v in "" ${VARIANTS:-}; do
  if [ -z "$v" ]; then unset GANGFIT_LIB; else export GANGFIT_LIB=$ROOT/k8s-spark-scheduler_amd/variants/libgangfit_$v.so; fi
  timeout 300 python tools/probe_variants.py chain 2>&1 | tail -1
done | tee "$OUT/variants.txt"
unset GANGFIT_LIB
timeout 200 python tools/probe_solo_phases.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/solo_phases.txt"
