#!/bin/bash
# PMC passes on the two quad-major fine-level launches + the GPU tests added since the last full run
mkdir -p gpurun_out/r03
R=${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests/test_model_harness.py -x -q -m gpu -k "quirk or whole_forward or other_sizes" > gpurun_out/r03/t_harness.txt 2>&1
tail -4 gpurun_out/r03/t_harness.txt
bash tools/pmc_one.sh tools/fine_only.py 0 r03_fineL0 fine_quad 0 > gpurun_out/r03/pmc_fineL0.txt 2>&1
bash tools/pmc_one.sh tools/fine_only.py 0 r03_fineL1 fine_quad 1 > gpurun_out/r03/pmc_fineL1.txt 2>&1
cat gpurun_out/r03/pmc_fineL0.txt gpurun_out/r03/pmc_fineL1.txt
