#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r03
cd /tmp && export TMPDIR=/tmp
run() {  # tag, env..., args
  tag=$1; shift
  env "$@" timeout 100 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $R/gpurun_out/r03/x_$tag -- python $R/tools/fq_exp.py $SIDE $BB 2 2>/dev/null | grep side
  python $R/tools/pmc_summary.py $(ls $R/gpurun_out/r03/x_$tag/*/*counter_collection.csv | head -1) | grep fine_quad
}
SIDE=104 BB=1 run b1 X=1
SIDE=104 BB=2 run b2 X=1
SIDE=104 BB=8 run b8 X=1
SIDE=80 BB=8 run s80 X=1
SIDE=64 BB=8 run s64 X=1
SIDE=128 BB=8 run s128 X=1
SIDE=104 BB=8 run nt1 CASMTR_FQ_FLAGS=1
SIDE=104 BB=8 run nt3 CASMTR_FQ_FLAGS=3
SIDE=104 BB=8 run sc1 CASMTR_FQ_FLAGS=5
SIDE=104 BB=8 run w320 CASMTR_FQ_WAVES_PER_XCD=320
SIDE=104 BB=8 run w256 CASMTR_FQ_WAVES_PER_XCD=256
cd $R
for f in 0 1 3 5; do CASMTR_FQ_FLAGS=$f python tools/fq_exp.py 104 8 20; done
