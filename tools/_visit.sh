R=$(pwd); mkdir -p $R/gpurun_out/r06; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/r06/kt_step7 -o k -- python $R/tools/run_kernel.py step7 30 > /dev/null 2>&1
cd $R; grep -E "roi_fwd_order|dma_inl_step" gpurun_out/r06/kt_step7/k_kernel_stats.csv | awk -F'",' '{print substr($1,1,70), $2, $3, $4}'
for i in 1 2 3; do timeout 600 python bench.py --no-cpu-baseline --no-e2e --no-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['launch_ms'])"; done
