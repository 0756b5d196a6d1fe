R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
timeout 1200 python -m pytest tests/test_gpu_ops.py -x -q -k "fine or qtatt or quad or level" > $O/r05v_t_ops.txt 2>&1; tail -n 2 $O/r05v_t_ops.txt
python tools/fq_samepair.py > $O/r05v_fq_samepair.txt 2>&1; cat $O/r05v_fq_samepair.txt
for i in 1 2; do
python bench.py --steps 80 --warmup 6 --no-extra --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('warm', d['value'], d['ms_per_step'], d['kernels']['qta_fine_level[lists<=64]']['ms_per_step'], d['kernels']['qta_fine_level[lists>64]']['ms_per_step'], d['kernels']['dual_softmax_fix']['ms_per_step'])"
CASMTR_FQ_FLAGS=16 python bench.py --steps 80 --warmup 6 --no-extra --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('nowarm', d['value'], d['ms_per_step'], d['kernels']['qta_fine_level[lists<=64]']['ms_per_step'], d['kernels']['qta_fine_level[lists>64]']['ms_per_step'])"
done
