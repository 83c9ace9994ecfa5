#!/bin/bash
# Config 5 on the GPU box: img/s of both variants at both score thresholds, the equivalence check, and the share of
# GPU time spent in tvmi:: kernels (rocprofv3 --kernel-trace --stats; kernel-trace only, no counters).
#   gpurun -- 'bash tools/e2e_profile.sh <tag>'
TAG=${1:-r02_e2e}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOTDIR=$(pwd)
timeout 600 python tools/e2e_maskrcnn.py --variant check > $OUT/check.json 2> $OUT/check.err; cat $OUT/check.json
for v in reference fused; do
  for t in 0.0 0.05; do
    timeout 600 python tools/e2e_maskrcnn.py --variant $v --score-thresh $t --steps 10 > $OUT/e2e_${v}_t$t.json 2> $OUT/e2e_${v}_t$t.err
    cat $OUT/e2e_${v}_t$t.json
  done
  cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $ROOTDIR/$OUT/prof_$v -o e2e -- python $ROOTDIR/tools/e2e_maskrcnn.py --variant $v --score-thresh 0.0 --steps 6 --warmup 3 > $ROOTDIR/$OUT/prof_$v.log 2>&1
  cd $ROOTDIR
  python tools/kernel_share.py $OUT/prof_$v > $OUT/kernel_share_$v.txt 2>&1; head -30 $OUT/kernel_share_$v.txt
done
