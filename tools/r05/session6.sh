R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r05h_prof -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra > /dev/null 2>&1
f=$(ls $O/r05h_prof/*/*kernel_stats.csv | head -1); cp $f $O/r05h_kernel_stats.csv; rm -rf $O/r05h_prof
head -45 $O/r05h_kernel_stats.csv | cut -c1-150
