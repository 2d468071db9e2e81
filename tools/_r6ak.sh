set -u
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=16
OUT=gpurun_out/$1; mkdir -p $OUT
export GANGFIT_LIB=$PWD/k8s-spark-scheduler_amd/variants/libgangfit_fillprobe.so
timeout 300 python tools/probe_minfrag.py azmajor 2>&1 | grep -v amdgpu.ids | head -4 > $OUT/fillprobe.txt
cat $OUT/fillprobe.txt
