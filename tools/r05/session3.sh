R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
$R/tools/probes/bin/gather_attrib > $O/r05c_gather_attrib.txt 2>&1
$R/tools/probes/bin/dma_width > $O/r05c_dma_width.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_ds_split.py -x -q > $O/r05c_t_split.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "dual_softmax" > $O/r05c_t_ops.txt 2>&1
CASMTR_DS_DEBUG=1 timeout 600 python bench.py --steps 6 --warmup 3 --no-extra --no-cpu-baseline > /dev/null 2> $O/r05c_dsdebug.txt
timeout 900 python bench.py --steps 40 --warmup 6 --no-extra --no-cpu-baseline > $O/r05c_bench.json 2> $O/r05c_bench.err
cat $O/r05c_gather_attrib.txt $O/r05c_dma_width.txt; tail -n 3 $O/r05c_t_split.txt $O/r05c_t_ops.txt; grep ds_xdecide $O/r05c_dsdebug.txt | sort | uniq -c | head; head -c 300 $O/r05c_bench.json
