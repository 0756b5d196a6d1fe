cd /tmp && export TMPDIR=/tmp
R=/root/repo
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAIT_INST_ANY" \
           "TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_MISC SQ_WAIT_ANY SQ_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_fine/p$i -- python $R/tools/fine_level_only.py 2 > /dev/null 2>&1
  f=$(ls $R/gpurun_out/pmc_fine/p$i/*/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python $R/tools/pmc_summary.py $f | grep quad_attn
done
