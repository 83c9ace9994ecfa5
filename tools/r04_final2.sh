#!/bin/bash
# Last GPU-box visit of round 4 (after the deform_conv2d backward work): the artefact round (tools/gpu_round.sh: contract line,
# kernel-trace stats of the same command, config matrix incl. the backward routes, FETCH / WRITE PMC passes + calibration,
# kernel traces of the backward), the whole -m gpu suite, one fuzz seed, the backward check table, the LDS float-add probe and
# the grid-rounds table.   gpurun -- 'bash tools/r04_final2.sh <tag>'
TAG=${1:-r04b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
bash tools/gpu_round.sh $TAG > $OUT/round.log 2>&1; tail -14 $OUT/round.log
timeout 1500 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1; tail -4 $OUT/pytest_gpu.log
timeout 300 python tools/fuzz_gpu.py 43 > $OUT/fuzz.log 2>&1; tail -3 $OUT/fuzz.log
timeout 300 python tools/dcn_bwd_check.py $OUT/dcn_bwd_check.json > $OUT/dcn_bwd_check.log 2>&1; grep "c4 backward" $OUT/dcn_bwd_check.log
[ -x tools/probe/lds_atomic_rate ] || hipcc --offload-arch=gfx950 -O3 -o tools/probe/lds_atomic_rate tools/probe/lds_atomic_rate.hip > /dev/null 2>&1
timeout 60 tools/probe/lds_atomic_rate > $OUT/lds_atomic_rate.txt 2>&1; cat $OUT/lds_atomic_rate.txt
python tools/kt_rounds.py $OUT/kt_dcn_bwd/k_kernel_trace.csv > $OUT/rounds_dcn_bwd.txt 2>&1
