for v in base prio1 prio3 base prio1 prio3; do
  cp _variants/libtvmi_kernels_$v.so vision_amd/_lib/libtvmi_kernels.so
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-e2e --no-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench $v', d['value'], d['ms_per_step'], d.get('roofline',{}).get('launch_ms'))"
done
cp _variants/libtvmi_kernels_base.so vision_amd/_lib/libtvmi_kernels.so
