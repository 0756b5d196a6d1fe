#!/bin/bash
mkdir -p gpurun_out/r03
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "cascade_attn" > gpurun_out/r03/t_cas.txt 2>&1
tail -12 gpurun_out/r03/t_cas.txt
CASMTR_FQ_DEBUG=1 timeout 120 python tools/cascade_only.py 2 2>&1 | grep -v amdgpu.ids | tail -12
timeout 300 python bench.py --steps 100 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); print({k:v['ms_per_step'] for k,v in d['kernels'].items()})"
