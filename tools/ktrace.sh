#!/bin/bash
# kernel-trace stats for a run_kernel.py target: bash tools/ktrace.sh <which> <tag> [reps]
WHICH=${1:-roi7}; TAG=${2:-kt}; REPS=${3:-10}
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OUT -o kt -- python $ROOTDIR/tools/run_kernel.py $WHICH $REPS > $OUT/log.txt 2>&1
cd $ROOTDIR; ls $OUT; head -12 $OUT/kt_kernel_stats.csv | cut -c1-180
