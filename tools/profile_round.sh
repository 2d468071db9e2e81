#!/bin/bash
# Profiling recipe of one round (run on the MI355X box through gpurun; outputs under gpurun_out/prof_<tag>/).
#   1. rocprofv3 --kernel-trace --stats of the bench command (with extras: every chain kernel appears)  -> kernel durations
#   2. one rocprofv3 --pmc pass per counter set, kernel-trace only (never combined with sys/runtime/hip/hsa trace domains):
#        FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum            headline kernel only (--no-extras)
#        SQ set (waves, cycles, instructions, waits) and an LDS set     with extras: the FIFO chain kernels are covered
#   3. the whole Filter through the C++ mirror of the reference's interface (host_bench)
set -u
TAG=${1:-r2}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
# (--worker-sets 0: the headline through the launch path only — every batch is a dispatch row; the resident worker of the
#  independent batch is ONE long dispatch and gets a trace of its own below)
BENCH="python $ROOT/bench.py --steps 50 --warmup 5 --windows 3 --filter-calls 30 --no-cpu-baseline --worker-sets 0 ${BENCH_ARGS:-}"
rocprofv3 -L > "$OUT/counters_available.txt" 2>&1
timeout 400 rocprofv3 --kernel-trace --stats -T -f csv -d "$OUT/stats" -o stats -- $BENCH > "$OUT/stats.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -T -f csv -d "$OUT/stats_worker" -o stats -- python $ROOT/bench.py --steps 200 --warmup 5 --windows 3 --no-cpu-baseline --headline-only > "$OUT/stats_worker.log" 2>&1
# the headline alone: every fit_independent_kernel dispatch of this trace is a headline launch (its average duration is the
# figure bench.py's roofline.kernel_ms must agree with)
timeout 300 rocprofv3 --kernel-trace --stats -T -f csv -d "$OUT/stats_headline" -o stats -- $BENCH --headline-only > "$OUT/stats_headline.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -T -f csv -d "$OUT/pmc_fetch" -o pmc -- $BENCH --headline-only > "$OUT/pmc_fetch.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -T -f csv -d "$OUT/pmc_write" -o pmc -- $BENCH --headline-only > "$OUT/pmc_write.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -T -f csv -d "$OUT/pmc_l2" -o pmc -- $BENCH --headline-only > "$OUT/pmc_l2.log" 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -T -f csv -d "$OUT/pmc_sq" -o pmc -- $BENCH > "$OUT/pmc_sq.log" 2>&1
# the plain FIFO chain alone: every fit_fifo_solo_kernel launch of these passes is a full replay of the headline chain
# (999 earlier drivers + 1) -> instructions per application for bench.py's roofline.fifo_chain (profiles/pmc_chain.json)
CHAIN="python $ROOT/bench.py --steps 20 --warmup 5 --windows 3 --filter-calls 30 --no-cpu-baseline --no-extras --fifo-protocols cold --worker-sets 0"
timeout 300 rocprofv3 --kernel-trace --stats -T -f csv -d "$OUT/stats_chain" -o stats -- $CHAIN > "$OUT/stats_chain.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT -T -f csv -d "$OUT/chain_sq" -o pmc -- $CHAIN > "$OUT/chain_sq.log" 2>&1
# BASELINE config 3 alone (10 000 nodes x 10 000 apps, both packers): kernel durations, HBM traffic, L2 hits / misses
C3="python $ROOT/bench.py --config3-only"
timeout 300 rocprofv3 --kernel-trace --stats -T -f csv -d "$OUT/stats_config3" -o stats -- $C3 > "$OUT/stats_config3.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -T -f csv -d "$OUT/c3_fetch" -o pmc -- $C3 > "$OUT/c3_fetch.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -T -f csv -d "$OUT/c3_write" -o pmc -- $C3 > "$OUT/c3_write.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -T -f csv -d "$OUT/c3_l2" -o pmc -- $C3 > "$OUT/c3_l2.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY -T -f csv -d "$OUT/c3_sq" -o pmc -- $C3 > "$OUT/c3_sq.log" 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE -T -f csv -d "$OUT/pmc_lds" -o pmc -- $BENCH > "$OUT/pmc_lds.log" 2>&1
# the whole Filter through the C++ mirror of the reference's interface (three configurations; profiles/<tag>_host_filter.txt)
if [ -x "$ROOT/k8s-spark-scheduler_amd/host_bench" ]; then
  ( cd "$ROOT/k8s-spark-scheduler_amd" && timeout 100 ./host_bench 10000 1000 2000 tightly-pack; timeout 100 ./host_bench 10000 1000 2000 single-az-tightly-pack; timeout 100 ./host_bench 10000 1000 2000 single-az-minimal-fragmentation; timeout 200 ./host_bench 100000 1000 20000 tightly-pack ) > "$OUT/host_filter.txt" 2>&1
fi
find "$OUT" -name '*.csv' | head -40
grep -h '^{' "$OUT"/*.log | cut -c1-300 | head -8
