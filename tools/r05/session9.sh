R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
timeout 600 python -m pytest tests/test_gpu_ds_split.py -x -q > $O/r05q_t_split.txt 2>&1; tail -n 2 $O/r05q_t_split.txt
for i in 1 2; do
python bench.py --steps 80 --warmup 6 --no-extra --no-cpu-baseline > $O/r05q_bench_a$i.json 2>/dev/null
CASMTR_LIB_PATH=$R/casmtr_amd/lib_noslp/libcasmtr_hip.so python bench.py --steps 80 --warmup 6 --no-extra --no-cpu-baseline > $O/r05q_bench_b$i.json 2>/dev/null
done
python - <<'PY'
import json,os
R=os.environ.get('GRAFT_REPO_ROOT','/root/repo')
for n in ('a1','b1','a2','b2'):
    d=json.load(open(f'{R}/gpurun_out/r05q_bench_{n}.json'))
    print(n, d['value'], d['ms_per_step'], {k:round(v['ms_per_step'],3) for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['ms_per_step'])[:9]})
PY
