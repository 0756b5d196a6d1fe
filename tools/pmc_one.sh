# PMC passes over one single-kernel workload: bash tools/pmc_one.sh <script.py> <debug_flags> <tag> <kernel-name-pattern>
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
SCRIPT=$1; FLAGS=${2:-0}; TAG=${3:-full}; PAT=${4:-kernel}; EXTRA=${5:-}
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_IFETCH SQ_INSTS_SMEM SQ_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_FLAT SQ_BUSY_CU_CYCLES SQ_WAVE_DEP_WAIT SQ_WAIT_IFETCH SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_THREAD_CYCLES_VALU" \
           "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_F16 SQ_INSTS_VALU_TRANS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_GDS SQ_WAIT_INST_LDS SQ_INSTS_FLAT" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_${TAG}/p$i -- python $R/$SCRIPT 2 $FLAGS $EXTRA > /dev/null 2>&1
  f=$(ls $R/gpurun_out/pmc_${TAG}/p$i/*/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python $R/tools/pmc_summary.py $f | grep -i "$PAT"
done
