#!/bin/bash
mkdir -p gpurun_out/r03
timeout 300 python tools/fq_sweep.py 2>&1 | grep "default"
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r03/t_all2.txt 2>&1
tail -4 gpurun_out/r03/t_all2.txt
timeout 300 python bench.py --steps 100 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); print({k:v['ms_per_step'] for k,v in d['kernels'].items()})"
