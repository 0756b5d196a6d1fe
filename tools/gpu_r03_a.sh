#!/bin/bash
# round-3 iteration script: new fine-level kernel tests + short bench (run via gpurun)
mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "quad_major or qtattb_levels or dma_kernel_shapes or selection_ties" > gpurun_out/r03/t_ops.txt 2>&1
tail -15 gpurun_out/r03/t_ops.txt
timeout 600 python bench.py --steps 50 --warmup 5 --no-extra --no-cpu-baseline > gpurun_out/r03/bench1.json 2> gpurun_out/r03/bench1.err
tail -3 gpurun_out/r03/bench1.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03/bench1.json'))
print(d['value'], d['ms_per_step'], d.get('parity'))
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['ms_per_step']): print(f"{k:40s} {v}")
PY
