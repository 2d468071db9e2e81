set -u
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=16
OUT=gpurun_out/r6bf; mkdir -p $OUT
for v in default rh6 rh24 rh64 default; do
  if [ $v = default ]; then unset GANGFIT_LIB; else export GANGFIT_LIB=$GRAFT_REPO_ROOT/k8s-spark-scheduler_amd/variants/libgangfit_$v.so; fi
  echo "== $v" >> $OUT/variants.txt
  timeout 200 python tools/probe_zoned_parts.py minimal-fragmentation 2>&1 | grep "the batch\|one executor\|one application " >> $OUT/variants.txt
  PROBE_ORDER=scattered timeout 200 python tools/probe_zoned_parts.py minimal-fragmentation 2>&1 | grep "the batch" | sed 's/^/scattered order: /' >> $OUT/variants.txt
done
cat $OUT/variants.txt
