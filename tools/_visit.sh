for cfg in "0 100" "1 100" "1 200" "1 150" "0 100" "1 200" "1 300"; do
set -- $cfg
TVMI_SET_OPTIONS=roi_align.fold_first_round_pct=$2 timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-e2e --no-configs --roi-fold-order $1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fold=$1 pct=$2', d['value'], d['ms_per_step'], d.get('roofline',{}).get('launch_ms'))"
done
