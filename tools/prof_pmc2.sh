#!/bin/bash
# reduced PMC set (cache behaviour): bash tools/prof_pmc2.sh <which> <tag>
WHICH=${1:-roi7}; TAG=${2:-pmc}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
i=0
for C in "FETCH_SIZE GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_TA_BUSY_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --kernel-trace -f csv -d $OUT/p$i -o p -- python $ROOTDIR/tools/run_kernel.py $WHICH 4 > $OUT/p$i.log 2>&1
done
cd $ROOTDIR
