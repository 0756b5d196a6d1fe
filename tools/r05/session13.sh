R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
for cfg in "2 8 24" "8 8 24" "2 8 48" "2 8 104"; do set -- $cfg
echo "== B=$1 H=$2 side=$3"; FQ_B=$1 FQ_H=$2 FQ_SIDE=$3 CASMTR_LW_DEBUG=1 timeout 300 python tools/fq_lw.py 2>&1 | grep -v amdgpu.ids | grep "2 consumers per loader, resident\|fine_vs_kernel, resident\|^fine_lw: 1024" | sort -u | head -4
done
