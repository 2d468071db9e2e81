set -u
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=16
OUT=gpurun_out/r6u; mkdir -p $OUT
timeout 400 python -m pytest tests/test_gpu_worker.py -m gpu -q -x --timeout 120 > $OUT/pytest_worker.log 2>&1; echo "pytest worker rc=$?"; tail -5 $OUT/pytest_worker.log
echo "== record cache" | tee -a $OUT/worker_sets.txt
timeout 200 python tools/probe_worker_sets.py 11:21 14:16 2>&1 | grep -v amdgpu.ids | tee -a $OUT/worker_sets.txt
echo "== no record cache" | tee -a $OUT/worker_sets.txt
GANGFIT_LIB=$PWD/k8s-spark-scheduler_amd/variants/libgangfit_norc.so timeout 200 python tools/probe_worker_sets.py 11:21 2>&1 | grep -v amdgpu.ids | tee -a $OUT/worker_sets.txt
GANGFIT_LIB=$PWD/k8s-spark-scheduler_amd/variants/libgangfit_prof.so timeout 200 python tools/probe_worker_phases.py 2>&1 | grep -v amdgpu.ids | tee $OUT/worker_phases.txt
timeout 200 python tools/probe_worker_stress.py 2>&1 | grep -v amdgpu.ids | tail -3
