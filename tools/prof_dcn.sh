#!/bin/bash
# PMC passes for the fused deform_conv2d kernel: bash tools/prof_dcn.sh <tag>
TAG=${1:-pmc_dcn}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
i=0
for C in "GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD TA_TA_BUSY_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum FETCH_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --kernel-trace -f csv -d $OUT/p$i -o p -- python $ROOTDIR/tools/run_kernel.py dcn 4 > $OUT/p$i.log 2>&1
done
cd $ROOTDIR
python tools/pmc_summary.py $OUT 2>/dev/null | grep -A30 "dcn_fwd_mfma" | head -40
