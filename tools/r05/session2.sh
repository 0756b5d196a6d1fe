R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_ds_split.py -x -q > $O/r05b_t_split.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "dual_softmax or tie" > $O/r05b_t_ops.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_dist_nccl.py tests/test_tie_semantics.py -x -q > $O/r05b_t_pipe.txt 2>&1
timeout 900 python bench.py --steps 40 --warmup 6 > $O/r05b_bench.json 2> $O/r05b_bench.err
tail -5 $O/r05b_t_split.txt $O/r05b_t_ops.txt $O/r05b_t_pipe.txt; tail -5 $O/r05b_bench.err; head -c 1500 $O/r05b_bench.json
