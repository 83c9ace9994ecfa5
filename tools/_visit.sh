ROOTDIR=$(pwd); OUT=gpurun_out/r06g; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for K in step7 roi7 roi7cl bwd7 bwd14 nms100k; do
  for C in FETCH_SIZE WRITE_SIZE; do
    TVMI_TOOL_SERIALIZED_PROFILER=1 timeout 300 rocprofv3 --pmc $C --kernel-trace -f csv -d $ROOTDIR/$OUT/pmc_${K}_$C -o p -- python $ROOTDIR/tools/run_kernel.py $K 6 > $ROOTDIR/$OUT/pmc_${K}_$C.log 2>&1
  done
done
[ -x $ROOTDIR/tools/probe/fetch_calib ] || hipcc --offload-arch=gfx950 -O3 -o $ROOTDIR/tools/probe/fetch_calib $ROOTDIR/tools/probe/fetch_calib.hip > /dev/null 2>&1
TVMI_TOOL_SERIALIZED_PROFILER=1 timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d $ROOTDIR/$OUT/calib_fetch -o p -- $ROOTDIR/tools/probe/fetch_calib > $ROOTDIR/$OUT/calib_fetch.log 2>&1
TVMI_TOOL_SERIALIZED_PROFILER=1 timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -f csv -d $ROOTDIR/$OUT/calib_write -o p -- $ROOTDIR/tools/probe/fetch_calib > $ROOTDIR/$OUT/calib_write.log 2>&1
cd $ROOTDIR; python tools/pmc_traffic.py $OUT | head -4 | cut -c1-200
