R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
timeout 900 python -m pytest tests/test_gpu_ds_split.py -x -q 2>&1 | tail -4
timeout 600 python tools/ds_wide_time.py 2>&1 | grep -v amdgpu.ids | tee $O/r05x_ds_wide.txt
