"""Summarise rocprofv3 --pmc CSV passes: per-kernel mean of every counter (+ kernel-trace durations)."""
import csv, glob, os, sys, collections
root = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(root, "p*", "*counter_collection.csv"))):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if pat and pat not in k: continue
        vals[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
dur = collections.defaultdict(list)
for f in sorted(glob.glob(os.path.join(root, "p*", "*kernel_trace.csv"))):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if pat and pat not in k: continue
        dur[k].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
for k in vals:
    print("==", k[:110])
    if dur[k]: print(f"   duration_us(mean over {len(dur[k])}) = {sum(dur[k])/len(dur[k]):.1f}")
    for c, v in sorted(vals[k].items()):
        print(f"   {c:32s} {sum(v)/len(v):16.1f}   (n={len(v)})")
