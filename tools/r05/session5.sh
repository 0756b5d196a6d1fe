R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
python tools/fq_samepair.py > $O/r05g_fq_samepair.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r05g_ds_prof -- python $R/tools/ds_only.py 6 0 split > /dev/null 2>&1
f=$(ls $O/r05g_ds_prof/*/*kernel_stats.csv | head -1); cp $f $O/r05g_ds_kernel_stats.csv; rm -rf $O/r05g_ds_prof
cat $O/r05g_fq_samepair.txt; head -25 $O/r05g_ds_kernel_stats.csv | cut -c1-160
