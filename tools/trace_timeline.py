"""Condenses a rocprofv3 kernel trace (…_kernel_trace.csv) into a per-launch timeline of the LAST iteration of a
repeated op: start offset, duration, stream / queue, kernel name.  Usage: trace_timeline.py <dir> <first-kernel-substr>"""
import csv, glob, os, sys
d, first = sys.argv[1], sys.argv[2]
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
# last iteration = from the last "first kernel" whose predecessor is not the same kernel
it = [i for i in starts if i == 0 or first not in rows[i - 1]["Kernel_Name"]]
begin = it[-1]
t0 = int(rows[begin]["Start_Timestamp"])
sel = rows[begin:]
print(f"{len(sel)} launches, span {(max(int(r['End_Timestamp']) for r in sel) - t0) / 1e3:.1f} us")
agg = {}
for r in sel:
    k = r["Kernel_Name"].split("(")[0][-40:]
    a = agg.setdefault(k, [0, 0.0])
    a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for k, (c, t) in agg.items(): print(f"  {k:42s} x{c:4d}  total {t:9.1f} us")
for r in sel[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    print(f"  +{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} us  dur {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f}  q{r.get('Queue_Id', '?')}  {r['Kernel_Name'].split('(')[0][-38:]}")
