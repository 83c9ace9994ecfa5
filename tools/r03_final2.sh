#!/bin/bash
# Second (shorter) final visit of round 3 after the staged-store RoIAlign kernel and the NMS hand-offs: full -m gpu suite, the
# contract line, rocprofv3 kernel-trace stats of the same bench command, the config matrix, FETCH / WRITE passes of the two
# forward kernels, one fuzz seed.    gpurun -- 'bash tools/r03_final2.sh <tag>'
TAG=${1:-r03h}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; ROOTDIR=$(pwd)
timeout 2400 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
timeout 600 python bench.py --steps 50 --warmup 10 > $OUT/bench.json 2> $OUT/bench.err; cut -c1-300 $OUT/bench.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $ROOTDIR/$OUT/prof -o bench -- python $ROOTDIR/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e > $ROOTDIR/$OUT/prof.log 2>&1)
timeout 600 python tools/gpu_matrix.py $OUT/matrix.json > $OUT/matrix.log 2>&1; tail -2 $OUT/matrix.log
for K in roi7 roi7cl; do for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && TVMI_TOOL_SERIALIZED_PROFILER=1 timeout 300 rocprofv3 --pmc $C --kernel-trace -f csv -d $ROOTDIR/$OUT/pmc_${K}_$C -o p -- python $ROOTDIR/tools/run_kernel.py $K 6 > $ROOTDIR/$OUT/pmc_${K}_$C.log 2>&1)
done; done
python tools/pmc_traffic.py $OUT > $OUT/traffic_summary.txt 2>&1; grep "roi_align" $OUT/traffic_summary.txt
timeout 300 python tools/fuzz_gpu.py 51 > $OUT/fuzz.log 2>&1; tail -1 $OUT/fuzz.log
