R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
for f in 0 16 0 16; do echo "CASMTR_FQ_FLAGS=$f"; CASMTR_FQ_FLAGS=$f FQ_LW_SKIP=1 timeout 300 python tools/fq_lw.py 2>&1 | grep "fine_quad_kernel"; done
