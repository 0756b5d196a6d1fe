R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
timeout 2400 python -m pytest tests -x -q -m gpu > $O/r05t_gputests.txt 2>&1; tail -n 5 $O/r05t_gputests.txt
timeout 900 python bench.py > $O/r05t_bench.json 2> $O/r05t_bench.err; head -c 400 $O/r05t_bench.json; tail -3 $O/r05t_bench.err
