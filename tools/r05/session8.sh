R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
CASMTR_DS_DEBUG=1 timeout 600 python bench.py --steps 2 --warmup 2 --no-extra --no-cpu-baseline > /dev/null 2> $O/r05o_dsdebug.txt; grep ds_xdecide $O/r05o_dsdebug.txt | sort | uniq -c
