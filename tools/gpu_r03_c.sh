#!/bin/bash
mkdir -p gpurun_out/r03
timeout 1500 python -m pytest tests/test_gpu_pipeline.py -x -q -k "named_config_chain" > gpurun_out/r03/t_chain.txt 2>&1
tail -25 gpurun_out/r03/t_chain.txt
( time timeout 1200 python bench.py > gpurun_out/r03/bench3.json 2> gpurun_out/r03/bench3.err ) 2>&1 | grep real
tail -5 gpurun_out/r03/bench3.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03/bench3.json'))
print(d['value'], d['ms_per_step'], d.get('parity'))
print(json.dumps(d['roofline'])); print(json.dumps(d['rooflines_all']))
for k,v in d.get('other_configs',{}).items(): print(k, v['value'], v['ms_per_step'], v['roofline'], v['parity'])
print(d.get('with_conf_matrix')); print(d.get('two_batches_in_flight')); print(d.get('whole_model',{}).get('value')); print(d.get('cpu_baseline'))
PY
for ge in 1 8; do
CASMTR_FORCE_DIST=1 timeout 300 python bench.py --steps 150 --warmup 5 --no-extra --no-cpu-baseline --gather-every $ge > gpurun_out/r03/bench_forcedist_ge$ge.json 2> gpurun_out/r03/bench_forcedist_ge$ge.err
python -c "
import json; d=json.load(open('gpurun_out/r03/bench_forcedist_ge$ge.json')); print('forced 1-rank RCCL group, gather-every $ge:', d['value'], d['ms_per_step'])"
done
timeout 300 python bench.py --steps 150 --warmup 5 --no-extra --no-cpu-baseline > gpurun_out/r03/bench_nodist.json 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/r03/bench_nodist.json')); print('no process group:', d['value'], d['ms_per_step'])"
