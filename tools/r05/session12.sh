R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
FQ_LW_SKIP=1 timeout 300 python tools/fq_lw.py > $O/r05w_fq_vs.txt 2>&1; tail -8 $O/r05w_fq_vs.txt
CASMTR_VS_FLAGS=1 FQ_LW_SKIP=1 timeout 300 python tools/fq_lw.py 2>&1 | tail -4
CASMTR_VS_DEBUG=1 FQ_LW_SKIP=1 timeout 300 python tools/fq_lw.py 2>&1 | grep "^fine_vs:" | sort | uniq -c | sort -rn | awk '{ $1=""; print }' | sort -u -k3,3 | head -8
