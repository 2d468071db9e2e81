#!/bin/bash
# SQ counters of one independent batch alone: tools/micro/pmc_one_batch.sh <workload> <algo>  -> gpurun_out/pmc_<workload>_<algo>.txt
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
W=$1; A=$2; OUT=$ROOT/gpurun_out/pmc1_${W}_$A; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -T -f csv -d $OUT/a -o pmc -- python $ROOT/tools/micro/probe_one_batch.py $W $A > $OUT/a.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU -T -f csv -d $OUT/b -o pmc -- python $ROOT/tools/micro/probe_one_batch.py $W $A > $OUT/b.log 2>&1
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/*/*pmc_counter_collection.csv") + glob.glob("$OUT/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "fit_independent" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("$W algo $A:", {k: round(sum(v) / len(v), 1) for k, v in sorted(acc.items())})
PY
