R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
timeout 300 python tools/fq_ablate.py > $O/r05w_fq_ablate.txt 2>&1; grep -v amdgpu.ids $O/r05w_fq_ablate.txt
