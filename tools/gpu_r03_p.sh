#!/bin/bash
# final PMC passes on the two fine-level launches and the cascade launch (separate --pmc passes, tools/pmc_one.sh)
mkdir -p gpurun_out/r03
bash tools/pmc_one.sh tools/fine_only.py 0 r03p_fineL0 fine_quad 0 > gpurun_out/r03/pmc2_fineL0.txt 2>&1
bash tools/pmc_one.sh tools/fine_only.py 0 r03p_fineL1 fine_quad 1 > gpurun_out/r03/pmc2_fineL1.txt 2>&1
bash tools/pmc_one.sh tools/cascade_only.py 0 r03p_cas cascade_quad > gpurun_out/r03/pmc2_cas.txt 2>&1
cat gpurun_out/r03/pmc2_fineL0.txt gpurun_out/r03/pmc2_fineL1.txt gpurun_out/r03/pmc2_cas.txt
