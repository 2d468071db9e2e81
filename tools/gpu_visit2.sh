#!/bin/bash
# round 4, second GPU visit: validation of the delta checkpoints / self-announced completion / 8-wavefront min-frag chain / API split
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r4b
mkdir -p "$OUT"
cd "$ROOT"
export GPU_MAX_HW_QUEUES=16
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|Error" "$OUT/pytest_gpu.log" | tail -5
( cd k8s-spark-scheduler_amd && timeout 300 ./host_test gpu ) > "$OUT/host_test_gpu.log" 2>&1; echo "host_test rc=$?"; tail -2 "$OUT/host_test_gpu.log"
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?"; tail -1 "$OUT/smoke.log"
timeout 300 python tools/probe_variants.py chain 2>&1 | tail -1 | tee "$OUT/variants.txt"
timeout 300 python tools/probe_c5_filter.py 100 > "$OUT/c5_filter.json" 2> "$OUT/c5_filter.err"; cat "$OUT/c5_filter.json"
timeout 200 python tools/probe_c5_chain.py 2>&1 | tail -8 | tee "$OUT/c5_chain.txt"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_steps20.json" 2> "$OUT/bench_steps20.err"; echo "bench20 rc=$?"
python - <<'PY'
import json,os
j=json.load(open(os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/r4b/bench_steps20.json")))
rf=j["roofline"]
print("value",j["value"],"ms_per_step",j["ms_per_step"],"kernel",rf["kernel"],"kernel_ms",rf["kernel_ms"],"bound",rf["bound"],"frac",rf["frac"])
print("fractions",{k:round(v["frac"],4) for k,v in rf["fractions"].items()})
print("blocking",json.dumps(rf["regimes"]["blocking_call"]))
print("e2e",json.dumps(j.get("end_to_end",{}).get("through_the_resident_worker"))[:900])
ex=j.get("extras",{})
print("c5",json.dumps(ex.get("config5_100k_nodes_fifo"))[:700])
for k in ("single_az_tightly_pack","az_aware_tightly_pack","minimal_fragmentation","single_az_minimal_fragmentation"):
    print(k,ex.get(k,{}).get("fifo_filter_p50_ms"))
print("c3",{k:(v["decisions_per_s"],v["kernel_ms"],v["roofline"]["bound"],v["roofline"]["frac"]) for k,v in ex.get("config3_10k_nodes_x_10k_apps",{}).items()})
print("grp",json.dumps(j.get("node_sharded",{}).get("in_library_multi_device_context"))[:600])
print("err",ex.get("error"))
PY
