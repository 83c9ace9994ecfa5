#!/bin/bash
# One GPU-box visit of round 4: usage  gpurun -- 'bash tools/r04_visit.sh <tag> <steps...>'
# steps: routes | variants | sizes | overlay | bench | benchq | full | prof | nms | "<any shell command>"
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOTDIR=$(pwd)
for STEP in "$@"; do
  case $STEP in
    routes)   timeout 900 python -m pytest tests/test_gpu_roi_routes.py -x -q > $OUT/t_routes.log 2>&1; tail -25 $OUT/t_routes.log ;;
    variants) timeout 600 python tools/roi_variants.py $OUT/variants.json > $OUT/variants.log 2>&1; tail -20 $OUT/variants.log ;;
    sizes)    timeout 1500 python -m pytest tests/test_gpu_baseline_sizes.py -q > $OUT/t_sizes.log 2>&1; tail -25 $OUT/t_sizes.log ;;
    overlay)  timeout 1200 python -m pytest tests/test_overlay.py -q -m gpu > $OUT/t_overlay.log 2>&1; tail -25 $OUT/t_overlay.log ;;
    bench)    timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; cut -c1-1500 $OUT/bench.json; tail -5 $OUT/bench.err ;;
    benchq)   timeout 600 python bench.py --no-e2e > $OUT/bench.json 2> $OUT/bench.err; cut -c1-1500 $OUT/bench.json; tail -5 $OUT/bench.err ;;
    full)     timeout 2400 python -m pytest tests -q -m gpu -x > $OUT/t_full.log 2>&1; tail -25 $OUT/t_full.log ;;
    prof)     cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $ROOTDIR/$OUT/prof -o bench -- python $ROOTDIR/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e > $ROOTDIR/$OUT/prof.log 2>&1; cd $ROOTDIR; python tools/kernel_share.py $OUT/prof 2>/dev/null | head -40 ;;
    nms)      timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_sizes.py -q -k nms -x > $OUT/t_nms.log 2>&1; tail -8 $OUT/t_nms.log
              timeout 600 python tools/nms_variants.py $OUT/nms_variants.json > $OUT/nms_variants.log 2>&1; tail -40 $OUT/nms_variants.log
              for K in nms100k nms100k_dense; do
                (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $ROOTDIR/$OUT/kt_$K -o k -- python $ROOTDIR/tools/run_kernel.py $K 6 > /dev/null 2>&1)
                python tools/nms_timeline.py $OUT/kt_$K 200 > $OUT/timeline_$K.txt 2>&1; head -16 $OUT/timeline_$K.txt
              done ;;
    *)        bash -c "$STEP" ;;
  esac
done
