R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_gpu_ops.py -x -q -k "fine or qtatt or quad or level" > $O/r05f_t_ops.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_ds_split.py -x -q > $O/r05f_t_split.txt 2>&1
python tools/fq_sweep.py > $O/r05f_fq_sweep.txt 2>&1
CASMTR_DS_DEBUG=1 timeout 600 python bench.py --steps 3 --warmup 2 --no-extra --no-cpu-baseline > /dev/null 2> $O/r05f_dsdebug.txt; grep ds_xdecide $O/r05f_dsdebug.txt | sort | uniq -c | head -3
timeout 900 python bench.py --steps 60 --warmup 6 --no-extra --no-cpu-baseline > $O/r05f_bench.json 2> $O/r05f_bench.err
tail -n 4 $O/r05f_t_ops.txt $O/r05f_t_split.txt; cat $O/r05f_fq_sweep.txt; python - <<'PY'
import json,os
d=json.load(open(os.environ.get('GRAFT_REPO_ROOT','/root/repo')+'/gpurun_out/r05f_bench.json'))
print(d['value'], d['ms_per_step'])
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['ms_per_step'])[:10]: print(k, v['ms_per_step'])
PY
