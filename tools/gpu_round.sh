#!/bin/bash
# One GPU-box visit: parity suite, smoke, bench line, rocprofv3 kernel-trace stats of the same bench command,
# PMC traffic pass for the dominant kernel.   Usage: gpurun -- 'bash tools/gpu_round.sh <tag>'
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOTDIR=$(pwd)
echo "== host: $(nproc) cpus; $(rocminfo 2>/dev/null | grep -m1 gfx9)"> $OUT/env.txt
python -c "import torch; print(torch.__version__, torch.version.hip, torch.cuda.get_device_name(0), torch.cuda.device_count())" >> $OUT/env.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -6 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
timeout 600 python bench.py --steps 50 --warmup 10 > $OUT/bench.json 2> $OUT/bench.err; cat $OUT/bench.json
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $ROOTDIR/$OUT/prof -o bench -- python $ROOTDIR/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $ROOTDIR/$OUT/prof.log 2>&1
cd $ROOTDIR
head -16 $OUT/prof/bench_kernel_stats.csv | cut -c1-160
# HBM traffic of the dominant kernel: separate --pmc passes (FETCH_SIZE and WRITE_SIZE cannot share a pass)
cd /tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d $ROOTDIR/$OUT/pmc_fetch -o p -- python $ROOTDIR/tools/run_kernel.py roi7 6 > $ROOTDIR/$OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -f csv -d $ROOTDIR/$OUT/pmc_write -o p -- python $ROOTDIR/tools/run_kernel.py roi7 6 > $ROOTDIR/$OUT/pmc_write.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d $ROOTDIR/$OUT/pmc_fetch_cl -o p -- python $ROOTDIR/tools/run_kernel.py roi7cl 6 > $ROOTDIR/$OUT/pmc_fetch_cl.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -f csv -d $ROOTDIR/$OUT/pmc_write_cl -o p -- python $ROOTDIR/tools/run_kernel.py roi7cl 6 > $ROOTDIR/$OUT/pmc_write_cl.log 2>&1
cd $ROOTDIR
# probe binaries are not tracked: build the calibration probe here if it did not travel with the snapshot
[ -x $ROOTDIR/tools/probe/fetch_calib ] || hipcc --offload-arch=gfx950 -O3 -o $ROOTDIR/tools/probe/fetch_calib $ROOTDIR/tools/probe/fetch_calib.hip > /dev/null 2>&1
# FETCH_SIZE / WRITE_SIZE calibration on known byte counts (guide: gfx950 FETCH_SIZE halves wide coalesced reads)
cd /tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d $ROOTDIR/$OUT/calib_fetch -o p -- $ROOTDIR/tools/probe/fetch_calib > $ROOTDIR/$OUT/calib_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -f csv -d $ROOTDIR/$OUT/calib_write -o p -- $ROOTDIR/tools/probe/fetch_calib > $ROOTDIR/$OUT/calib_write.log 2>&1
cd $ROOTDIR
grep -h calib $OUT/calib_fetch/*counter_collection.csv | cut -d, -f 1-3,10- | head; grep -h calib $OUT/calib_write/*counter_collection.csv | head -4
