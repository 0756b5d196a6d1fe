# Round-5 counter set for the gather kernels (VERDICT r04 "missing" item 3): bash tools/r05/pmc_gather.sh <script.py> <tag> <kernel-pattern> [script args...]
# Separate --pmc passes (never combined with tracing); counters absent from `rocprofv3 -L` on this box are dropped from a pass.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
SCRIPT=$1; TAG=$2; PAT=$3; shift 3
AVAIL=$R/gpurun_out/r05_counters.txt
[ -s $AVAIL ] || rocprofv3 -L > $AVAIL 2>&1
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_IFETCH SQ_INSTS_SMEM SQ_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_FLAT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CU_CYCLES" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum TA_FLAT_WAVEFRONTS_sum" \
           "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum TD_SPI_STALL_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_TAG_STALL_sum TCC_BUSY_sum"; do
  i=$((i+1))
  use=""
  for c in $set; do grep -q -w "$c" $AVAIL && use="$use $c"; done
  [ -z "$use" ] && continue
  timeout 150 rocprofv3 --pmc $use --output-format csv -d $R/gpurun_out/pmc_${TAG}/p$i -- python $R/$SCRIPT "$@" > /dev/null 2>&1
  f=$(ls $R/gpurun_out/pmc_${TAG}/p$i/*/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python $R/tools/pmc_summary.py $f | grep -i "$PAT"
done
