set -u
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=16
OUT=gpurun_out/r6ar; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_worker.py tests/test_gpu_zones.py tests/test_gpu_fullsize.py tests/test_gpu_feasible.py tests/test_gpu_sharded.py -m gpu -q -x --timeout 600 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $OUT/pytest.log | tail -2
for lib in "" noahead "" noahead; do
  if [ -n "$lib" ]; then export GANGFIT_LIB=$PWD/k8s-spark-scheduler_amd/variants/libgangfit_$lib.so; else unset GANGFIT_LIB; fi
  timeout 300 python tools/probe_variants.py ind 2>&1 | grep -v amdgpu.ids >> $OUT/variants.txt
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']['regimes']
print('${lib:-default}', 'value', d['value'], 'kernel_ms', d['roofline']['kernel_ms'], 'launch', r['launch_per_batch'], 'congested', r['congested']['value'], r['congested']['kernel_ms'])
" >> $OUT/bench_ab.txt
done
cat $OUT/variants.txt; cat $OUT/bench_ab.txt
