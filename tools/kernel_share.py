"""Share of GPU time per kernel family from a rocprofv3 --kernel-trace --stats output directory:
ours (namespace tvmi::) vs library kernels (MIOpen / hipBLASLt / ATen)."""
import csv
import glob
import os
import sys

d = sys.argv[1]
files = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
if not files:
    sys.exit(f"no *kernel_stats.csv under {d}")
rows = list(csv.DictReader(open(files[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
ours = [r for r in rows if "tvmi::" in r["Name"]]
t_ours = sum(float(r["TotalDurationNs"]) for r in ours)
print(f"kernels: {len(rows)}   GPU time in kernels: {tot / 1e6:.2f} ms   in tvmi:: kernels: {t_ours / 1e6:.2f} ms = {100 * t_ours / tot:.1f} %")
print("top 12 kernels:")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:12]:
    print(f"  {100 * float(r['TotalDurationNs']) / tot:5.1f} %  calls {r['Calls']:>6}  avg {float(r['AverageNs']) / 1e3:9.1f} us  {r['Name'][:110]}")
print("tvmi:: kernels:")
for r in sorted(ours, key=lambda r: -float(r["TotalDurationNs"])):
    print(f"  {100 * float(r['TotalDurationNs']) / tot:5.2f} %  calls {r['Calls']:>6}  avg {float(r['AverageNs']) / 1e3:9.1f} us  {r['Name'][:110]}")
