#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r03
# cascade-shaped problem through the fine-level kernel: 208x208, H = 4, 25 parents (100 candidates), no top-k
python tools/fq_exp.py 208 8 20 4 25
CASMTR_FQ_WAVES_PER_XCD=512 python tools/fq_exp.py 208 8 20 4 25
CASMTR_FQ_WAVES_PER_XCD=256 python tools/fq_exp.py 208 8 20 4 25
python tools/fq_exp.py 416 8 10 2 25
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03/prof_h -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra > /dev/null 2>&1
f=$(find $R/gpurun_out/r03/prof_h -name "*kernel_stats.csv" | head -1); head -25 $f | cut -c1-160
