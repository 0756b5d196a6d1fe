# FETCH_SIZE / WRITE_SIZE of cascade_quad_kernel per window field and schedule: bash tools/pmc_cq_fields.sh  -> gpurun_out/pmc_cq_fields.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_cq_fields.txt
: > $OUT
for field in smooth mix12 random; do
  for dyn in 0 1; do
    for ctr in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
      d=$R/gpurun_out/pmc_cq/${field}_${dyn}_$(echo $ctr | tr ' ' '_')
      timeout 200 rocprofv3 --pmc $ctr --output-format csv -d $d -- python $R/tools/cq_fields.py one $field $dyn > /dev/null 2>&1
      f=$(ls $d/*/*counter_collection.csv 2>/dev/null | head -1)
      [ -n "$f" ] && echo "== $field dynamic=$dyn $ctr" >> $OUT && python $R/tools/pmc_summary.py $f | grep -i "cascade_quad" >> $OUT
    done
  done
done
cat $OUT
