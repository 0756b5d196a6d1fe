R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
timeout 300 python tools/fq_lw.py > $O/r05w_fq_lw.txt 2>&1; grep -v "^fine_lw:" $O/r05w_fq_lw.txt | tail -8
CASMTR_LW_DEBUG=1 timeout 300 python tools/fq_lw.py 2>&1 | grep "^fine_lw:" | sort | uniq -c | sort -rn | awk '{ $1=""; print }' | sort -u -k3,3 | head -8
