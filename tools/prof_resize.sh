#!/bin/bash
# HBM-traffic and issue counters of the tiled resize kernels: bash tools/prof_resize.sh <tag>
TAG=${1:-pmc_resize}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
for K in resize_bilinear resize_bilinear_aa resize_bicubic; do
  i=0
  for C in "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
           "TA_TA_BUSY_sum TD_TD_BUSY_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS"; do
    i=$((i+1))
    timeout 200 rocprofv3 --pmc $C --kernel-trace -f csv -d $OUT/$K/p$i -o p -- python $ROOTDIR/tools/run_kernel.py $K 4 > $OUT/${K}_p$i.log 2>&1
  done
  python $ROOTDIR/tools/pmc_summary.py $OUT/$K tile_kernel > $OUT/$K.txt 2>&1
done
cd $ROOTDIR; cat $OUT/*.txt
