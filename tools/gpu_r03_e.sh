#!/bin/bash
mkdir -p gpurun_out/r03
R=${GRAFT_REPO_ROOT:-/root/repo}
timeout 300 python tools/fq_sweep.py 2>&1 | grep level
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "qtattb_levels or dma_kernel_shapes or selection_ties" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
for lv in 0 1; do
timeout 150 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $R/gpurun_out/r03/pmc_tcc_L$lv -- python $R/tools/fine_only.py 2 0 $lv > /dev/null 2>&1
python $R/tools/pmc_summary.py $(ls $R/gpurun_out/r03/pmc_tcc_L$lv/*/*counter_collection.csv | head -1) | grep fine_quad
timeout 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r03/pmc_fetch_L$lv -- python $R/tools/fine_only.py 2 0 $lv > /dev/null 2>&1
python $R/tools/pmc_summary.py $(ls $R/gpurun_out/r03/pmc_fetch_L$lv/*/*counter_collection.csv | head -1) | grep fine_quad
timeout 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r03/pmc_write_L$lv -- python $R/tools/fine_only.py 2 0 $lv > /dev/null 2>&1
python $R/tools/pmc_summary.py $(ls $R/gpurun_out/r03/pmc_write_L$lv/*/*counter_collection.csv | head -1) | grep fine_quad
done
timeout 150 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $R/gpurun_out/r03/pmc_tcc_probe -- $R/tools/probes/bin/gather_quad > /dev/null 2>&1
python $R/tools/pmc_summary.py $(ls $R/gpurun_out/r03/pmc_tcc_probe/*/*counter_collection.csv | head -1)
cd $R
timeout 300 python bench.py --steps 100 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); print({k:v['ms_per_step'] for k,v in d['kernels'].items()})"
