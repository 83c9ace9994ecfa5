#!/bin/bash
# One GPU-box visit that produces every artefact quoted in DESIGN.md §6 (copied from gpurun_out/<tag>/ to profiles/):
# bench line, rocprofv3 kernel-trace stats of the same bench command, the measured config matrix, HBM-traffic PMC
# passes (FETCH_SIZE / WRITE_SIZE in SEPARATE passes, kernel-trace only) for the dominant forward kernel, the
# channels_last kernel, the tile-owner backward and the large-NMS mask kernel, and the FETCH/WRITE calibration probe.
#   gpurun -- 'bash tools/gpu_round.sh <tag>'
TAG=${1:-r06}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOTDIR=$(pwd)
echo "== host: $(nproc) cpus; $(rocminfo 2>/dev/null | grep -m1 gfx9)"> $OUT/env.txt
python -c "import torch; print(torch.__version__, torch.version.hip, torch.cuda.get_device_name(0), torch.cuda.device_count())" >> $OUT/env.txt 2>&1
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; cut -c1-400 $OUT/bench.json
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $ROOTDIR/$OUT/prof -o bench -- python $ROOTDIR/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-configs > $ROOTDIR/$OUT/prof.log 2>&1
cd $ROOTDIR
timeout 600 python tools/gpu_matrix.py $OUT/matrix.json > $OUT/matrix.log 2>&1; tail -3 $OUT/matrix.log
cd /tmp
for K in ${TVMI_ROUND_PMC:-step7 roi7 roi7cl bwd7 bwd14 nms100k}; do
  for C in FETCH_SIZE WRITE_SIZE; do
    TVMI_TOOL_SERIALIZED_PROFILER=1 timeout 300 rocprofv3 --pmc $C --kernel-trace -f csv -d $ROOTDIR/$OUT/pmc_${K}_$C -o p -- python $ROOTDIR/tools/run_kernel.py $K 6 > $ROOTDIR/$OUT/pmc_${K}_$C.log 2>&1
  done
done
for K in ${TVMI_ROUND_KT:-bwd7:20 bwd14:20 nms100k:10 dcn_bwd:10 dcn_bwd_dw:10}; do
  timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $ROOTDIR/$OUT/kt_${K%%:*} -o k -- python $ROOTDIR/tools/run_kernel.py ${K%%:*} ${K##*:} > /dev/null 2>&1
done
cd $ROOTDIR
# probe binaries are not tracked: build the calibration probe here if it did not travel with the snapshot
[ -x $ROOTDIR/tools/probe/fetch_calib ] || hipcc --offload-arch=gfx950 -O3 -o $ROOTDIR/tools/probe/fetch_calib $ROOTDIR/tools/probe/fetch_calib.hip > /dev/null 2>&1
cd /tmp
TVMI_TOOL_SERIALIZED_PROFILER=1 timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d $ROOTDIR/$OUT/calib_fetch -o p -- $ROOTDIR/tools/probe/fetch_calib > $ROOTDIR/$OUT/calib_fetch.log 2>&1
TVMI_TOOL_SERIALIZED_PROFILER=1 timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -f csv -d $ROOTDIR/$OUT/calib_write -o p -- $ROOTDIR/tools/probe/fetch_calib > $ROOTDIR/$OUT/calib_write.log 2>&1
cd $ROOTDIR
python tools/pmc_traffic.py $OUT > $OUT/traffic_summary.txt 2>&1; cat $OUT/traffic_summary.txt
