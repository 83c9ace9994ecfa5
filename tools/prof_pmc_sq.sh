#!/bin/bash
# SQ-only PMC passes (issue / wait / instruction mix) for one kernel driver: bash tools/prof_pmc_sq.sh <which> <tag>
WHICH=${1:-nms100k}; TAG=${2:-pmc_sq}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
i=0
[ -n "$PMC_ONLY_TAIL" ] && SKIP=3 || SKIP=0
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
         "SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" \
         "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_F32 SQ_VALU_MFMA_BUSY_CYCLES" \
         "TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum" "FETCH_SIZE" "WRITE_SIZE" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum" "SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_ANY SQ_WAVES"; do
  i=$((i+1))
  [ $i -le $SKIP ] && continue
  TVMI_TOOL_SERIALIZED_PROFILER=1 timeout 300 rocprofv3 --pmc $C --kernel-trace -f csv -d $OUT/p$i -o p -- python $ROOTDIR/tools/run_kernel.py $WHICH 3 > $OUT/p$i.log 2>&1
done
cd $ROOTDIR
python tools/pmc_summary.py $OUT ${3:-} > $OUT/summary.txt 2>&1
