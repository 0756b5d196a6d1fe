R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_ds_split.py -x -q > $O/r05u_t_split.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "dual_softmax" > $O/r05u_t_ops.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r05u_prof -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra > /dev/null 2>&1
f=$(ls $O/r05u_prof/*/*kernel_stats.csv | head -1); cp $f $O/r05u_kernel_stats.csv; rm -rf $O/r05u_prof
tail -n 3 $O/r05u_t_split.txt $O/r05u_t_ops.txt; grep -E "ds_x|ds_fix" $O/r05u_kernel_stats.csv | sed 's/(.*)"/"/' | cut -d, -f1-4
cd $R; python tools/fq_sweep.py 2>&1 | grep -E "default|320|512" | head -8
