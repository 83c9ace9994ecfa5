"""Per-launch timeline of the LAST torchvision::nms call in a rocprofv3 kernel trace of tools/run_kernel.py nms100k[_dense].
Usage: nms_timeline.py <trace dir> [max rows]"""
import csv, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
NAMES = ("score_keys", "widen_index", "nms_push", "nms_mask_tiles", "nms_resolve_wide", "nms_colreduce", "nms_survivor_offsets", "nms_compact_order", "nms_sweep_small",
         "fillBuffer", "copyBuffer", "radix_sort", "merge", "transform", "fill_reverse")
def nm(r):
    for t in NAMES:
        if t in r["Kernel_Name"]:
            return t
    return r["Kernel_Name"][:30]
# a call starts with the score sort: its first kernel is score_keys (score_sort.hip) or, for other score types, torch.sort's prologue
cut = max(i for i, r in enumerate(rows) if nm(r) in ("score_keys", "fill_reverse"))
sel = rows[cut:]
t0 = int(sel[0]["Start_Timestamp"])
print(f"{len(sel)} launches, span {(max(int(r['End_Timestamp']) for r in sel) - t0) / 1e3:.1f} us")
agg = {}
for r in sel:
    a = agg.setdefault(nm(r), [0, 0.0]); a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for k, (c, t) in agg.items():
    print(f"  {k:24s} x{c:4d}  total {t:9.1f} us")
for r in sel[: int(sys.argv[2]) if len(sys.argv) > 2 else 200]:
    print(f"  +{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} us  dur {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f}  q{r['Queue_Id']}  {nm(r)}")
