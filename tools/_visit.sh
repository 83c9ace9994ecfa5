R=$(pwd); OUT=$R/gpurun_out/r06_icache; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
i=0
for C in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQC_TC_INST_REQ SQC_TC_DATA_READ_REQ SQC_TC_STALL SQC_ICACHE_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU"; do
  i=$((i+1))
  for K in roi7 step7; do
  TVMI_TOOL_SERIALIZED_PROFILER=1 timeout 300 rocprofv3 --pmc $C --kernel-trace -f csv -d $OUT/${K}_p$i -o p -- python $R/tools/run_kernel.py $K 3 > $OUT/${K}_p$i.log 2>&1
  done
done
cd $R
for K in roi7 step7; do mkdir -p $OUT/$K; for d in $OUT/${K}_p*; do [ -d $d ] && mv $d $OUT/$K/$(basename $d | sed "s/${K}_//"); done; python tools/pmc_summary.py $OUT/$K roi_align_fwd_ms_dma_inl > $OUT/${K}_summary.txt 2>&1; cat $OUT/${K}_summary.txt; done
