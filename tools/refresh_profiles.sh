#!/bin/bash
# Round-end measurement set (run on the GPU box via gpurun): bench lines, rocprofv3 kernel stats, FETCH/WRITE PMC passes.
# Usage: bash tools/refresh_profiles.sh r02a   -> files under gpurun_out/<tag>_*; copy the ones to keep into profiles/.
TAG=${1:-rXX}
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out
mkdir -p $O
cd $R
python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err                                     # headline: configs[1], 250 steps (+ other_configs legs, whole model)
python bench.py --config 2c --steps 100 --warmup 5 > $O/${TAG}_bench_2c.json 2>> $O/${TAG}_bench.err        # configs[3]
python bench.py --config indoor --steps 300 --warmup 10 > $O/${TAG}_bench_indoor.json 2>> $O/${TAG}_bench.err  # configs[4] shapes
python bench.py --masked --steps 100 --warmup 5 --no-extra --no-cpu-baseline > $O/${TAG}_bench_masked.json 2>> $O/${TAG}_bench.err   # configs[2] shapes on 1 GPU
python bench.py --steps 100 --warmup 5 --no-extra --no-cpu-baseline --shared-inputs --explicit-windows > $O/${TAG}_bench_r01_dataflow.json 2>> $O/${TAG}_bench.err
python bench.py --steps 50 --warmup 5 --no-extra --with-callers > $O/${TAG}_bench_callers.json 2>> $O/${TAG}_bench.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof_2c -- python $R/bench.py --config 2c --steps 5 --warmup 2 --no-cpu-baseline --no-extra > /dev/null 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_pmc_fetch -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra > /dev/null 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_pmc_write -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof_callers -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra --with-callers > /dev/null 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_pmc_fetch_callers -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra --with-callers > /dev/null 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_pmc_write_callers -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra --with-callers > /dev/null 2>&1
find $O -name "*kernel_stats.csv" -newer $O/${TAG}_bench.json | head
find $O -name "*counter_collection.csv" -newer $O/${TAG}_bench.json | head
head -c 600 $O/${TAG}_bench.json
