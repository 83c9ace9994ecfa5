#!/bin/bash
# Final GPU-box visit of round 3: the artefact round (tools/gpu_round.sh), the whole -m gpu suite, two fuzz seeds, SQ / TA PMC
# passes of the three deform_conv2d kernels, the large-NMS variants + timelines.   gpurun -- 'bash tools/r03_final.sh <tag>'
TAG=${1:-r03b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
bash tools/gpu_round.sh $TAG > $OUT/round.log 2>&1; tail -12 $OUT/round.log
timeout 2400 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1; tail -5 $OUT/pytest_gpu.log
for SEED in 31 32; do timeout 300 python tools/fuzz_gpu.py $SEED >> $OUT/fuzz.log 2>&1; done; tail -3 $OUT/fuzz.log
for K in dcn dcn_bf16 dcn_dw; do bash tools/prof_pmc_sq.sh $K $TAG/pmc_$K dcn_fwd > /dev/null 2>&1; head -3 $OUT/pmc_$K/summary.txt; done
timeout 600 python tools/nms_variants.py $OUT/nms_variants.json > $OUT/nms_variants.log 2>&1; tail -8 $OUT/nms_variants.log
for K in nms100k nms100k_dense; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/$OUT/ktl_$K -o k -- python $GRAFT_REPO_ROOT/tools/run_kernel.py $K 6 > /dev/null 2>&1)
  python tools/nms_timeline.py $OUT/ktl_$K 200 > $OUT/timeline_$K.txt 2>&1; head -3 $OUT/timeline_$K.txt
done
