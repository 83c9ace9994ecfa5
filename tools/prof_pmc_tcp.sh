#!/bin/bash
# Texture-path PMC passes (TLB, TCP stalls, latencies) for one kernel driver: bash tools/prof_pmc_tcp.sh <which|probe> <tag> [kernel-name filter]
# "probe" profiles tools/probe/dma_sector (the same DMA instruction stream at the rate the texture path sustains).
WHICH=${1:-roi7}; TAG=${2:-pmc_tcp}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
if [ "$WHICH" = probe ]; then CMD="$ROOTDIR/tools/probe/dma_sector"; else CMD="python $ROOTDIR/tools/run_kernel.py $WHICH 3"; fi
i=0
for C in "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum" \
         "TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_THRASHING_STALL_sum" \
         "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum" \
         "TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum" \
         "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TA_TCP_STATE_READ_sum" \
         "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TC_STALL_sum" \
         "TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_TCR_RDRET_STALL_sum TD_SPI_STALL_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum" \
         "TCC_BUSY_sum TCC_TAG_STALL_sum TCC_IB_STALL_sum TCC_REQ_sum" \
         "TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_WAVES"; do
  i=$((i+1))
  TVMI_TOOL_SERIALIZED_PROFILER=1 timeout 300 rocprofv3 --pmc $C --kernel-trace -f csv -d $OUT/p$i -o p -- $CMD > $OUT/p$i.log 2>&1
done
cd $ROOTDIR
python tools/pmc_summary.py $OUT ${3:-} > $OUT/summary.txt 2>&1
