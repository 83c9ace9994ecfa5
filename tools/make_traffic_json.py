"""profiles/roofline_traffic.json from a gpu_round.sh output directory (FETCH_SIZE / WRITE_SIZE passes)."""
import csv, glob, json, os, sys
tag_dir = sys.argv[1]
fetch_factor = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0   # gfx950 FETCH_SIZE calibration (tools/probe/fetch_calib)
write_factor = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
def mean_counter(sub, counter, kernel_pat):
    vals = []
    for f in glob.glob(os.path.join(tag_dir, sub, "*counter_collection.csv")):
        for row in csv.DictReader(open(f)):
            if kernel_pat in row["Kernel_Name"] and row["Counter_Name"] == counter:
                vals.append(float(row["Counter_Value"]))
    return sum(vals) / len(vals) if vals else None
out = {}
for key, pat, fsub, wsub in (("roi_align_fwd_ms_dma", "roi_align_fwd_ms_dma", "pmc_fetch", "pmc_write"),
                             ("roi_align_fwd_nhwc", "roi_align_fwd_nhwc", "pmc_fetch_cl", "pmc_write_cl")):
    f = mean_counter(fsub, "FETCH_SIZE", pat); w = mean_counter(wsub, "WRITE_SIZE", pat)
    if f is None or w is None: continue
    out[key] = {"fetch_size_kb": f, "write_size_kb": w, "fetch_factor": fetch_factor, "write_factor": write_factor,
                "hbm_bytes_per_launch": int(f * 1024 * fetch_factor + w * 1024 * write_factor),
                "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, KB units, per launch mean; factors from tools/probe/fetch_calib"}
json.dump(out, open(os.path.join("profiles", "roofline_traffic.json"), "w"), indent=1)
print(out)
