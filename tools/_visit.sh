mkdir -p gpurun_out/r06
export TMPDIR=/tmp; R=$(pwd); cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/r06/kt_plane -o k -- python $R/bench.py --no-cpu-baseline --no-e2e --no-configs --roi-plane 1 --steps 20 --warmup 5 > $R/gpurun_out/r06/kt_plane.log 2>&1
python - $R/gpurun_out/r06/kt_plane <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print(f"  {r['Name'][:80]:80s} calls {r['Calls']:>4} avg_us {float(r['AverageNs'])/1e3:8.1f} min {float(r['MinNs'])/1e3:8.1f} max {float(r['MaxNs'])/1e3:8.1f}")
PY
