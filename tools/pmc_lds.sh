cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for spec in "tools/fine_only.py 2 0 0:fine_quad" "tools/fine_only.py 2 0 1:fine_quad" "tools/cascade_only.py 2 0:cascade_quad"; do
  args=${spec%%:*}; pat=${spec##*:}
  timeout 150 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/pmc_lds/$pat$RANDOM -- python $R/$args > /dev/null 2>&1
done
for f in $R/gpurun_out/pmc_lds/*/*/*counter_collection.csv; do python $R/tools/pmc_summary.py $f | grep -i "fine_quad\|cascade_quad" | cut -c1-400; done
