#!/bin/bash
# MFMA utilisation of the deform_conv2d calls of BASELINE config 4 (north_star: "rocprof MFMA util reported"): one SQ-counter
# pass per call shape (counters only + kernel-trace, as the pool requires), summarised by tools/mfma_busy.py into
# profiles/dcn_mfma_busy.json, which bench.py attaches to the `configs.deform_conv2d_*` rows of the contract line.
#   gpurun -- 'bash tools/mfma_busy.sh r06'
TAG=${1:-r06}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG/mfma; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
for W in dcn dcn_bf16 dcn_bwd dcn_bwd_bf16; do
  TVMI_TOOL_SERIALIZED_PROFILER=1 timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY \
    --kernel-trace -f csv -d $OUT/$W -o p -- python $ROOTDIR/tools/run_kernel.py $W 4 > $OUT/$W.log 2>&1
done
cd $ROOTDIR
python tools/mfma_busy.py $OUT $OUT/dcn_mfma_busy.json; cat $OUT/dcn_mfma_busy.json
