#!/usr/bin/env python3
"""HBM traffic per launch from the FETCH_SIZE / WRITE_SIZE passes of tools/gpu_round.sh.

    python tools/pmc_traffic.py gpurun_out/<tag> [--write-json]

Reads <dir>/pmc_<case>_<COUNTER>/**/*counter_collection.csv (one counter per pass, kernel-trace only), takes the mean
over the launches of each kernel of interest (the first launch of a case is dropped: cold caches), converts KB -> bytes
and applies the gfx950 corrections:
  * FETCH_SIZE: MI355X_MICROARCH.md (HBM section) — the counter tallies 128-byte requests at 64 bytes, i.e. reports
    exactly 1/2 of a wide coalesced streaming read; accesses that are whole >=128-byte runs (LDS-DMA rows, 16 B/lane
    loads) get factor 2.0.  Kernels whose reads are a mix of 64- and 128-byte requests get the factor measured by
    tools/probe/fetch_calib on that access shape (calib_fetch/ in the same directory, printed below).
  * WRITE_SIZE: uncalibrated in the guide; fetch_calib's streaming write reports 1.0x, used as is.
With --write-json the forward kernels' figures go to profiles/roofline_traffic.json (bench.py reads `traffic` there)."""
import csv
import glob
import json
import os
import sys

# case -> [(kernel name pattern, fetch factor, algorithmic note)]
CASES = {
    "roi7": [("roi_align_fwd_ms_dma", 1.33)],
    "step7": [("roi_align_fwd_ms_dma_inl_step", 1.33)],
    "roi7cl": [("roi_align_fwd_nhwc", 2.0)],
    "bwd7": [("roi_align_bwd_owner<float, 7", 2.0), ("roi_bwd_prepass", 2.0)],
    "bwd14": [("roi_align_bwd_owner<float, 14", 2.0), ("roi_bwd_prepass", 2.0)],
    "nms100k": [("nms_mask_tiles", 2.0), ("nms_resolve_wide", 2.0), ("nms_colreduce", 2.0)],
}


def per_kernel(dirname, counter):
    vals = {}
    for f in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter:
                vals.setdefault(row["Kernel_Name"], []).append(float(row["Counter_Value"]))
    return vals


def mean_for(vals, pat):
    sel = [(k, v) for k, v in vals.items() if pat in k]
    if not sel:
        return None, 0
    allv = []
    for _, v in sel:
        allv += v[1:] if len(v) > 1 else v
    return sum(allv) / len(allv), len(allv)


def main():
    d = sys.argv[1]
    out = {}
    for case, kernels in CASES.items():
        fv = per_kernel(os.path.join(d, f"pmc_{case}_FETCH_SIZE"), "FETCH_SIZE")
        wv = per_kernel(os.path.join(d, f"pmc_{case}_WRITE_SIZE"), "WRITE_SIZE")
        for pat, ff in kernels:
            f, nf = mean_for(fv, pat)
            w, nw = mean_for(wv, pat)
            if f is None or w is None:
                print(f"{case:8s} {pat:28s} (no samples)")
                continue
            rd, wr = f * 1024 * ff, w * 1024
            print(f"{case:8s} {pat:28s} launches {nf:3d}/{nw:3d}  FETCH_SIZE {f:12.1f} KB x{ff:.2f} = {rd / 1e6:9.2f} MB   "
                  f"WRITE_SIZE {w:12.1f} KB = {wr / 1e6:9.2f} MB   total {(rd + wr) / 1e6:9.2f} MB")
            out[f"{case}:{pat}"] = {"fetch_size_kb": f, "write_size_kb": w, "fetch_factor": ff, "write_factor": 1.0,
                                    "hbm_bytes_per_launch": int(rd + wr)}
    for sub, counter in (("calib_fetch", "FETCH_SIZE"), ("calib_write", "WRITE_SIZE")):
        for k, v in sorted(per_kernel(os.path.join(d, sub), counter).items()):
            print(f"calib    {k[:60]:60s} {counter} {sum(v) / len(v):12.1f} KB")
    if "--write-json" in sys.argv:
        note = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (kernel-trace only), KB units, per-launch mean; "
                "fetch factor per tools/pmc_traffic.py")
        js = {}
        for key, src in (("roi_align_fwd_ms_dma", "roi7:roi_align_fwd_ms_dma"), ("roi_align_fwd_ms_dma_inl_step", "step7:roi_align_fwd_ms_dma_inl_step"), ("roi_align_fwd_nhwc", "roi7cl:roi_align_fwd_nhwc"),
                         ("roi_align_bwd_owner_7", "bwd7:roi_align_bwd_owner<float, 7"), ("roi_align_bwd_owner_14", "bwd14:roi_align_bwd_owner<float, 14"),
                         ("nms_mask_tiles_100k", "nms100k:nms_mask_tiles")):
            if src in out:
                js[key] = dict(out[src], note=note)
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        # fingerprint of the kernel sources the counters were taken on: bench.py reports whether the shipped sources still
        # match (the traffic figure is a committed measurement, not a constant of nature)
        import hashlib
        h = hashlib.sha256()
        for name in ("roi_align.hip", "roi_common.h"):
            h.update(open(os.path.join(root, "vision_amd", "csrc", name), "rb").read())
        js["_measured_on"] = {"roi_align_sources_sha16": h.hexdigest()[:16], "directory": os.path.basename(os.path.normpath(d))}
        json.dump(js, open(os.path.join(root, "profiles", "roofline_traffic.json"), "w"), indent=1)
        print("wrote profiles/roofline_traffic.json")


if __name__ == "__main__":
    main()
