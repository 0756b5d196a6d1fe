R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
fail=0
for i in $(seq 1 25); do
  timeout 300 python -m pytest tests/test_gpu_fine_variants.py -x -q 2>&1 | tail -1 | grep -q "10 passed" || { fail=$((fail+1)); echo "run $i failed"; }
done
echo "variant test: $fail failures in 25 runs"
for i in 1 2 3; do FQ_B=8 timeout 300 python tools/fq_lw.py 2>&1 | grep -c "bit-equal True"; done
