#!/bin/bash
mkdir -p gpurun_out/r03
R=${GRAFT_REPO_ROOT:-/root/repo}
CASMTR_FQ_DEBUG=1 timeout 60 python tools/fine_only.py 1 0 0 2>&1 | grep fine_quad | head -2
CASMTR_FQ_DEBUG=1 timeout 60 python tools/fine_only.py 1 0 1 2>&1 | grep fine_quad | head -2
timeout 300 python tools/fq_sweep.py 2>&1 | grep level
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "qtattb_levels or dma_kernel_shapes or selection_ties" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
for lv in 0 1; do
timeout 150 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $R/gpurun_out/r03/pmc_tcc2_L$lv -- python $R/tools/fine_only.py 2 0 $lv > /dev/null 2>&1
python $R/tools/pmc_summary.py $(ls $R/gpurun_out/r03/pmc_tcc2_L$lv/*/*counter_collection.csv | head -1) | grep fine_quad
done
cd $R
timeout 300 python bench.py --steps 100 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); print({k:v['ms_per_step'] for k,v in d['kernels'].items()})"
