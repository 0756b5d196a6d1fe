R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
for route in quads tokens quads tokens; do
CASMTR_CALLER_LAYOUT=$route timeout 600 python bench.py --steps 30 --warmup 5 --no-extra --with-callers > $O/r05z_callers_$route.json 2> $O/r05z_callers_$route.err
python - $route <<'PY'
import json,sys
d=json.loads(open(f'gpurun_out/r05z_callers_{sys.argv[1]}.json').read().strip().splitlines()[-1])
k=d['kernels']
print(sys.argv[1], d['value'], d['ms_per_step'], {n:(round(v['ms_per_step'],2),v.get('launches_per_step')) for n,v in k.items() if isinstance(v,dict) and v.get('ms_per_step',0)>0.3})
PY
done
