#!/bin/bash
# Round-end measurement set (run on the GPU box via gpurun): bench lines, rocprofv3 kernel stats, FETCH/WRITE PMC passes.
# Usage: bash tools/refresh_profiles.sh r01d   -> files under gpurun_out/<tag>_*; copy the ones to keep into profiles/.
TAG=${1:-rXX}
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out
mkdir -p $O
cd $R
python bench.py --steps 10 --warmup 3 > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --materialize-conf > $O/${TAG}_bench_materialized_conf.json 2>> $O/${TAG}_bench.err
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --channels-last > $O/${TAG}_bench_channels_last.json 2>> $O/${TAG}_bench.err
python bench.py --steps 10 --warmup 3 --with-callers > $O/${TAG}_bench_callers.json 2>> $O/${TAG}_bench.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof_callers -- python $R/bench.py --steps 5 --warmup 2 --with-callers > /dev/null 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_pmc_fetch -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_pmc_write -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
find $O -name "*kernel_stats.csv" -newer $O/${TAG}_bench.json | head
find $O -name "*counter_collection.csv" -newer $O/${TAG}_bench.json | head
head -c 600 $O/${TAG}_bench.json
