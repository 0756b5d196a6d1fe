R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
python bench.py > $O/r05c_bench.json 2> $O/r05c_bench.err
python bench.py --steps 50 --warmup 5 --no-extra --with-callers > $O/r05c_bench_callers.json 2>> $O/r05c_bench.err
python - <<'PY'
import json
for n in ("bench","bench_callers"):
    d=json.loads(open(f'gpurun_out/r05c_{n}.json').read().strip().splitlines()[-1])
    print(n, d['value'], d['ms_per_step'], (d.get('whole_model') or {}).get('value'))
PY
