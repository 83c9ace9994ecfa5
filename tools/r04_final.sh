#!/bin/bash
# Final GPU-box visit of round 4: the artefact round (tools/gpu_round.sh: contract line, kernel-trace stats of the same command,
# config matrix, FETCH / WRITE PMC passes + calibration), the whole -m gpu suite, two fuzz seeds, SQ / texture-path PMC passes of
# the dominant kernel, the forward launch-route table.   gpurun -- 'bash tools/r04_final.sh <tag>'
TAG=${1:-r04}
OUT=gpurun_out/$TAG
mkdir -p $OUT
bash tools/gpu_round.sh $TAG > $OUT/round.log 2>&1; tail -14 $OUT/round.log
timeout 2400 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1; tail -4 $OUT/pytest_gpu.log
for SEED in 41 42; do timeout 300 python tools/fuzz_gpu.py $SEED >> $OUT/fuzz.log 2>&1; done; tail -3 $OUT/fuzz.log
bash tools/prof_pmc.sh roi7 $TAG/pmc_sq_roi7 > /dev/null 2>&1; python tools/pmc_summary.py $OUT/pmc_sq_roi7 > $OUT/pmc_sq_roi7_summary.txt 2>&1; grep -A40 "roi_align_fwd_ms_dma" $OUT/pmc_sq_roi7_summary.txt | head -44
timeout 600 python tools/roi_variants.py $OUT/roi_variants.json > $OUT/roi_variants.log 2>&1; grep -v amdgpu $OUT/roi_variants.log | head -12
