#!/bin/bash
# One GPU visit: the -m gpu parity suite, the torch-free host tests, the driver's bench command and the default one, then
# the profiling recipe.  Everything lands under gpurun_out/<tag>/ (merged back by gpurun).
#   tools/gpu_round.sh <tag> [go|tests|bench|two|zoned|c5|variants|phases|feasible|mf|prof ...]   (default: tests bench prof)
#     go    is there a Go toolchain on the box (integration/go/run_pins.sh needs Go 1.19)?
#     two   `python bench.py --gpus 2` on this one-GPU box: the N > 1 control flow starting its own ranks
set -u
TAG=${1:-r5}
shift || true
WHAT=${*:-tests bench prof}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
export GPU_MAX_HW_QUEUES=16
for w in $WHAT; do
  case $w in
    tests)
      timeout 1500 python -m pytest tests -m gpu -q --timeout 900 --durations=15 > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/pytest_gpu.log"
      grep -E "passed|failed" "$OUT/pytest_gpu.log" | tail -3
      ( cd k8s-spark-scheduler_amd && timeout 300 ./host_test gpu ) > "$OUT/host_test_gpu.log" 2>&1; echo "host_test rc=$?" | tee -a "$OUT/host_test_gpu.log"
      tail -2 "$OUT/host_test_gpu.log"
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?"; tail -1 "$OUT/smoke.log"
      ;;
    go)
      { command -v go; ls -d /usr/local/go /usr/lib/go* 2>/dev/null; go version 2>&1; } > "$OUT/go_probe.txt" 2>&1; cat "$OUT/go_probe.txt"
      # (run_pins.sh also needs a checkout of the reference, which does not travel to the box: the probe only records
      #  whether the toolchain exists there)
      ;;
    bench)
      t0=$(date +%s.%N)
      GANGFIT_BENCH_FULL="$OUT/bench_full_steps20.json" timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_steps20.json" 2> "$OUT/bench_steps20.err"; echo "bench20 rc=$? wall=$(python -c "import time,sys; print(round(time.time()-float(sys.argv[1]),1))" $t0) s"
      tail -1 "$OUT/bench_steps20.json" | wc -c; tail -1 "$OUT/bench_steps20.json"
      GANGFIT_BENCH_FULL="$OUT/bench_full_default_headline.json" timeout 900 python bench.py --no-extras --no-cpu-baseline > "$OUT/bench_default_headline.json" 2> "$OUT/bench_default.err"; echo "bench rc=$?"
      tail -1 "$OUT/bench_default_headline.json" | cut -c1-600
      ;;
    two)
      GANGFIT_BENCH_FULL="$OUT/bench_full_two_ranks.json" bash tools/smoke_two_ranks_one_gpu.sh > "$OUT/two_ranks.log" 2>&1; cat "$OUT/two_ranks.log"
      cp /tmp/two_ranks.out "$OUT/two_ranks.out" 2>/dev/null; tail -30 /tmp/two_ranks.err > "$OUT/two_ranks.err" 2>/dev/null
      ;;
    zoned)  # the zone-aware packers' independent batch: parity subset, then the four-kernel path against the one-launch path
      timeout 900 python -m pytest tests/test_gpu_zones.py tests/test_gpu_minfrag.py tests/test_gpu_fullsize.py tests/test_host_mirror.py -m gpu -q --timeout 600 > "$OUT/pytest_zoned.log" 2>&1; echo "pytest zoned rc=$?"; tail -3 "$OUT/pytest_zoned.log"
      timeout 300 python tools/probe_zoned_batch.py > "$OUT/zoned_batch.txt" 2>&1; echo "zoned probe rc=$?"; cat "$OUT/zoned_batch.txt"
      ;;
    c5)  # config 5: the chain per rotated head, fast and slow cluster
      timeout 600 python tools/probe_c5_heads.py 100 "$OUT/c5_heads.json" > "$OUT/c5_heads.txt" 2>&1; echo "c5 rc=$?"; cat "$OUT/c5_heads.txt"
      ;;
    variants)  # cold chains of every packer on the build variants under k8s-spark-scheduler_amd/variants/ (tools/build_variant.sh)
      timeout 300 python tools/probe_variants.py chain > "$OUT/variants.txt" 2>&1
      for v in k8s-spark-scheduler_amd/variants/libgangfit_*.so; do
        [ -f "$v" ] && GANGFIT_LIB=$PWD/$v timeout 300 python tools/probe_variants.py chain >> "$OUT/variants.txt" 2>&1
      done
      cat "$OUT/variants.txt"
      ;;
    phases)  # where the chain kernels' cycles go (instrumented variants), per application
      { timeout 300 python tools/probe_solo_phases.py; timeout 300 python tools/probe_zoned.py 10000 3 azmajor; timeout 300 python tools/probe_minfrag.py azmajor; } > "$OUT/chain_phases.txt" 2>&1; echo "phases rc=$?"; cat "$OUT/chain_phases.txt"
      ;;
    feasible)  # the lone blocking call: gf_fit_batch against gf_fit_feasible
      timeout 300 python tools/probe_feasible.py > "$OUT/feasible_call.txt" 2>&1; echo "feasible rc=$?"; cat "$OUT/feasible_call.txt"
      ;;
    mf)  # minimal-fragmentation chains: parity subset, phases, cold chain times of every packer
      timeout 900 python -m pytest tests/test_gpu_minfrag.py tests/test_gpu_feasible.py tests/test_gpu_incremental.py tests/test_gpu_zones.py -m gpu -q -x --timeout 600 > "$OUT/pytest_mf.log" 2>&1; echo "pytest mf rc=$?"; tail -3 "$OUT/pytest_mf.log"
      timeout 300 python tools/probe_minfrag.py azmajor > "$OUT/minfrag_phases.txt" 2>&1; cat "$OUT/minfrag_phases.txt"
      timeout 300 python tools/probe_variants.py chain > "$OUT/chain_times.txt" 2>&1; cat "$OUT/chain_times.txt"
      ;;
    prof)
      bash tools/profile_round.sh "$TAG" ${PROF_GROUPS:-} > "$OUT/profile.log" 2>&1; tail -12 "$OUT/profile.log"
      ;;
  esac
done
