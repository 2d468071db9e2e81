#!/bin/bash
# Profiling recipe of one round (run on the MI355X box through gpurun; outputs under gpurun_out/prof_<tag>/).
#   tools/profile_round.sh <tag> [headline chain config3 zoned full host ...]      (default: all six groups)
# Every group = one bench.py command profiled by SEPARATE rocprofv3 runs: --kernel-trace --stats for the durations, then one
# --pmc run per counter set with --kernel-trace only (never combined with sys / runtime / hip / hsa trace domains):
#   FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum | the SQ set (waves, cycles, instructions by type, waits)
#   headline  bench.py --headline-only at the DRIVER's --steps 20 --warmup 5 (hl20) and at bench.py's defaults (hl, K = 2 000): both
#             regimes of the headline in one process — the launch path's windows are rows of fit_independent_kernel (one dispatch
#             per batch), every worker window is ONE dispatch of fit_worker_kernel that serves the window's K tickets (median
#             over the dispatches / K = per ticket)                                      -> profiles/pmc_headline.json (runs.stepsK)
#   chain     bench.py --no-extras --fifo-protocols cold: every fit_fifo_solo_kernel launch replays the headline chain
#                                                                                        -> profiles/pmc_chain.json
#   config3   bench.py --config3-only (10 000 nodes x 10 000 apps, both packers)          -> profiles/pmc_config3.json
#   zoned     the kernels of the zone-aware and minimal-fragmentation packers, ONE instantiation per profiled command
#             (tools/profile_cmd.py): the one-launch independent batch (fit_zoned_fused_kernel: single-az-tightly-pack,
#             single-az-minimal-fragmentation), the LDS chains (fit_fifo_zoned_lds_kernel: single-az / az-aware tightly-pack;
#             fit_fifo_minfrag_lds_kernel: plain = the 8-wavefront instantiation, single-AZ = the 16-wavefront one); the
#             four counter sets and the LDS set each                                  -> profiles/pmc_zoned.json
#   full      the driver's command with extras (every chain kernel appears): durations only + the LDS counter set
#   host      the whole Filter through the C++ mirror of the reference's interface (host_bench) -> profiles/<tag>_host_filter.txt
set -u
TAG=${1:-r5}
shift || true
GROUPS_=${*:-headline chain config3 zoned full host}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
SQSET="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_ANY"
run_group() {  # <prefix> <timeout> <command...>: stats + the four counter passes of one command
  local pre=$1 to=$2; shift 2
  timeout $to rocprofv3 --kernel-trace --stats -T -f csv -d "$OUT/${pre}_stats" -o stats -- "$@" > "$OUT/${pre}_stats.log" 2>&1
  timeout $to rocprofv3 --kernel-trace --pmc FETCH_SIZE -T -f csv -d "$OUT/${pre}_fetch" -o pmc -- "$@" > "$OUT/${pre}_fetch.log" 2>&1
  timeout $to rocprofv3 --kernel-trace --pmc WRITE_SIZE -T -f csv -d "$OUT/${pre}_write" -o pmc -- "$@" > "$OUT/${pre}_write.log" 2>&1
  timeout $to rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -T -f csv -d "$OUT/${pre}_l2" -o pmc -- "$@" > "$OUT/${pre}_l2.log" 2>&1
  timeout $to rocprofv3 --kernel-trace --pmc $SQSET -T -f csv -d "$OUT/${pre}_sq" -o pmc -- "$@" > "$OUT/${pre}_sq.log" 2>&1
  # what the ALUs themselves report: the fraction of the kernel's cycles the vector / scalar ALUs process instructions
  # (rocprofv3's derived VALUBusy / SALUBusy) and the pipes' raw active cycles, the instruction fetches, the issue stalls
  if [ -n "${BUSY_PASS:-}" ]; then
    timeout $to rocprofv3 --kernel-trace --pmc VALUBusy SALUBusy -T -f csv -d "$OUT/${pre}_busy" -o pmc -- "$@" > "$OUT/${pre}_busy.log" 2>&1
    timeout $to rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_IFETCH SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_BUSY_CU_CYCLES -T -f csv -d "$OUT/${pre}_sq2" -o pmc -- "$@" > "$OUT/${pre}_sq2.log" 2>&1
  fi
}
for g in $GROUPS_; do
  case $g in
    headline)
      BUSY_PASS=1
      run_group hl20 300 python $ROOT/bench.py --headline-only --steps 20 --warmup 5 ${BENCH_ARGS:-}
      run_group hl 300 python $ROOT/bench.py --headline-only ${BENCH_ARGS:-}
      BUSY_PASS=
      ;;
    chain)
      run_group chain 300 python $ROOT/bench.py --steps 20 --warmup 5 --windows 3 --filter-calls 30 --no-cpu-baseline --no-extras --fifo-protocols cold --worker-sets 0
      ;;
    config3)
      BUSY_PASS=1
      run_group c3 300 python $ROOT/bench.py --config3-only
      BUSY_PASS=
      ;;
    zoned)
      LDSSET="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVES SQ_WAVE_CYCLES"
      for spec in "zb_saz batch single-az-tightly-pack" "zb_smf batch single-az-minimal-fragmentation" "zb_mf batch minimal-fragmentation" \
                  "zc_saz chain single-az-tightly-pack" "zc_aza chain az-aware-tightly-pack" \
                  "mc_mf chain minimal-fragmentation" "mc_smf chain single-az-minimal-fragmentation"; do
        set -- $spec
        # ZONED_SPECS="zb_smf ...": only these instantiations (tools/summarize_profile.py keeps the others' entries of pmc_zoned.json)
        if [ -n "${ZONED_SPECS:-}" ] && ! echo " $ZONED_SPECS " | grep -q " $1 "; then continue; fi
        run_group $1 120 python $ROOT/tools/profile_cmd.py $2 $3
        timeout 120 rocprofv3 --kernel-trace --pmc $LDSSET -T -f csv -d "$OUT/$1_lds" -o pmc -- python $ROOT/tools/profile_cmd.py $2 $3 > "$OUT/$1_lds.log" 2>&1
      done
      ;;
    full)
      FULL="python $ROOT/bench.py --steps 50 --warmup 5 --windows 3 --filter-calls 30 --no-cpu-baseline --worker-sets 0 ${BENCH_ARGS:-}"
      timeout 500 rocprofv3 --kernel-trace --stats -T -f csv -d "$OUT/stats" -o stats -- $FULL > "$OUT/stats.log" 2>&1
      timeout 500 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_WAVE_CYCLES -T -f csv -d "$OUT/pmc_lds" -o pmc -- $FULL > "$OUT/pmc_lds.log" 2>&1
      ;;
    host)
      if [ -x "$ROOT/k8s-spark-scheduler_amd/host_bench" ]; then
        ( cd "$ROOT/k8s-spark-scheduler_amd" && timeout 100 ./host_bench 10000 1000 2000 tightly-pack; timeout 100 ./host_bench 10000 1000 2000 single-az-tightly-pack; timeout 100 ./host_bench 10000 1000 2000 single-az-minimal-fragmentation; timeout 200 ./host_bench 100000 1000 20000 tightly-pack ) > "$OUT/host_filter.txt" 2>&1
      fi
      ;;
  esac
done
rocprofv3 -L > "$OUT/counters_available.txt" 2>&1
# summarised HERE (gpurun copies back 64 MiB at most, the raw CSVs of a K = 2 000 window are larger): the summaries go to
# $OUT/summary/ — copy them into profiles/ — and the raw pass directories are removed
cd "$ROOT" && python tools/summarize_profile.py "$TAG" "$OUT" "$OUT/summary" > "$OUT/summarize.log" 2>&1; echo "summarize rc=$?"; tail -3 "$OUT/summarize.log" | cut -c1-300
grep -h '^{' "$OUT"/*.log | cut -c1-300 | head -8
find "$OUT" -mindepth 1 -maxdepth 1 -type d ! -name summary -exec rm -rf {} +
du -sh "$OUT"
