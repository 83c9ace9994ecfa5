#!/bin/bash
# PMC passes for one kernel driver: bash tools/prof_pmc.sh <which> <tag>
WHICH=${1:-roi7}; TAG=${2:-pmc}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
i=0
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
         "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE" \
         "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum" \
         "TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_WAVEFRONTS_sum TD_TD_BUSY_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --kernel-trace -f csv -d $OUT/p$i -o p -- python $ROOTDIR/tools/run_kernel.py $WHICH 4 > $OUT/p$i.log 2>&1
done
cd $ROOTDIR
find $OUT -name "*counter_collection.csv" | head
