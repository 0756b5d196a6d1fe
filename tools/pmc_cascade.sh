# PMC passes over the cascade attention kernel alone: bash tools/pmc_cascade.sh <debug_flags> <tag>
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
FLAGS=${1:-0}
TAG=${2:-full}
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAIT_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_ANY SQ_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_SMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_IFETCH SQ_INSTS_WAVE32_LDS SQ_BUSY_CU_CYCLES"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_cas_$TAG/p$i -- python $R/tools/cascade_only.py 2 $FLAGS > /dev/null 2>&1
  f=$(ls $R/gpurun_out/pmc_cas_$TAG/p$i/*/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python $R/tools/pmc_summary.py $f | grep -i "cascade\|quad_attn"
done
