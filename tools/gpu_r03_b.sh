#!/bin/bash
# full GPU suite + bench with parity (run via gpurun)
mkdir -p gpurun_out/r03
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r03/t_all.txt 2>&1
tail -15 gpurun_out/r03/t_all.txt
timeout 900 python bench.py --steps 100 --warmup 5 > gpurun_out/r03/bench2.json 2> gpurun_out/r03/bench2.err
tail -3 gpurun_out/r03/bench2.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03/bench2.json'))
print(d['value'], d['ms_per_step'], d.get('parity'))
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['ms_per_step']): print(f"{k:40s} {v}")
PY
