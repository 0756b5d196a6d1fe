# first GPU call of round 5: counter list, attribution probe, baseline bench + sweep, PMC of the gather kernels
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
rocprofv3 -L > $O/r05_counters.txt 2>&1
(time $R/tools/probes/bin/gather_attrib) > $O/r05_gather_attrib.txt 2>&1
python bench.py --steps 60 --warmup 6 --no-extra --no-cpu-baseline > $O/r05a_bench.json 2> $O/r05a_bench.err
python tools/fq_sweep.py > $O/r05a_fq_sweep.txt 2>&1
bash tools/r05/pmc_gather.sh tools/fine_only.py r05_fineL0 fine_quad 2 0 0 > $O/r05_pmc_fineL0.txt 2>&1
bash tools/r05/pmc_gather.sh tools/fine_only.py r05_fineL1 fine_quad 2 0 1 > $O/r05_pmc_fineL1.txt 2>&1
bash tools/r05/pmc_gather.sh tools/cascade_only.py r05_cas cascade_quad 2 0 > $O/r05_pmc_cascade.txt 2>&1
rm -rf $O/pmc_r05_*
tail -40 $O/r05_gather_attrib.txt; head -c 700 $O/r05a_bench.json; echo; cat $O/r05a_fq_sweep.txt | tail -12
