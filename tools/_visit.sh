mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "nms_step" 2>&1 | tail -2
export TMPDIR=/tmp; R=$(pwd); cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/r06/kt_step_fused -o k -- python $R/tools/step_nms_trace.py fused 40 > $R/gpurun_out/r06/kt_step_fused.log 2>&1
python - $R/gpurun_out/r06/kt_step_fused <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:3]:
    print(f"  {r['Name'][:90]:90s} calls {r['Calls']:>4} avg_us {float(r['AverageNs'])/1e3:8.1f} min {float(r['MinNs'])/1e3:8.1f} max {float(r['MaxNs'])/1e3:8.1f}")
PY
