#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
for o in r c; do
echo "order $o"
CASMTR_CQ_ORDER=$o timeout 120 python tools/cascade_only.py 2 2>&1 | grep "pair kernel" | head -2
done
cd /tmp && export TMPDIR=/tmp
for o in r c; do
CASMTR_CQ_ORDER=$o timeout 150 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $R/gpurun_out/r03/pmc_cq_$o -- python $R/tools/cascade_only.py 2 > /dev/null 2>&1
python $R/tools/pmc_summary.py $(ls $R/gpurun_out/r03/pmc_cq_$o/*/*counter_collection.csv | head -1) | grep -i "cascade"
CASMTR_CQ_ORDER=$o timeout 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r03/pmc_cqf_$o -- python $R/tools/cascade_only.py 2 > /dev/null 2>&1
python $R/tools/pmc_summary.py $(ls $R/gpurun_out/r03/pmc_cqf_$o/*/*counter_collection.csv | head -1) | grep -i "cascade"
done
cd $R
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "cascade_attn or quad_major_layout" 2>&1 | tail -2
timeout 300 python bench.py --steps 100 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); print({k:v['ms_per_step'] for k,v in d['kernels'].items()})"
