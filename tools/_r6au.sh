set -u
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=16
OUT=gpurun_out/r6au; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_feasible.py tests/test_gpu_zones.py tests/test_host_mirror.py -m gpu -q -x --timeout 600 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|Error|assert" $OUT/pytest.log | tail -5
( cd k8s-spark-scheduler_amd && timeout 300 ./host_test gpu ) > $OUT/host_test_gpu.log 2>&1; echo "host_test rc=$?"; tail -1 $OUT/host_test_gpu.log
timeout 300 python tools/probe_feasible.py 2>&1 | grep -v amdgpu.ids > $OUT/feasible_call.txt; grep -E "single-az|az-aware" $OUT/feasible_call.txt | head -12
GANGFIT_LIB=$PWD/k8s-spark-scheduler_amd/variants/libgangfit_head.so timeout 300 python tools/probe_feasible.py 2>&1 | grep -v amdgpu.ids > $OUT/feasible_call_head.txt; echo "== head"; grep -E "single-az|az-aware" $OUT/feasible_call_head.txt | head -12
