#!/usr/bin/env python3
"""Times the multi-scale RoIAlign forward of BASELINE config 2 under the forward routes (tvmi_set_option):
per-roi (round-2 kernels only), planes (whole-plane levels staged, no bands), planes+bands, default (device-side decision),
for 7x7 / 14x14, fp32 / bf16.  4 rotated input sets, HIP events.  usage: roi_variants.py out.json"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vision_amd  # noqa: E402,F401
import bench  # noqa: E402

dev = torch.device("cuda")
sets = []
for i in range(bench.N_SETS):
    feats, boxes, _ = bench.make_inputs(dev, 1000 + 97 * i)
    from vision_amd.poolers import _convert_to_roi_format
    sets.append(dict(flist=[feats[str(j)] for j in range(4)], rois=_convert_to_roi_format(boxes).float()))
scales = [1.0 / s for s in bench.STRIDES]


def tm(fn, n=24, warm=4):
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


BASE = {"roi_align.shared_staging": 0, "roi_align.min_band_rows": 32, "roi_align.staging_gain_x16": 32,
        "roi_align.stage_whole_planes": 1, "roi_align.band_channels": 2}
FORCE = dict(BASE, **{"roi_align.shared_staging": 1, "roi_align.staging_gain_x16": 1 << 20})
ROUTES = {
    "per-roi": dict(BASE, **{"roi_align.shared_staging": 0}),
    "planes-only": dict(FORCE, **{"roi_align.min_band_rows": 0}),
    "bands1-only": dict(FORCE, **{"roi_align.stage_whole_planes": 0, "roi_align.band_channels": 1}),          # P2 + P3 as 1-channel bands
    "bands2-only": dict(FORCE, **{"roi_align.stage_whole_planes": 0, "roi_align.band_channels": 2, "roi_align.min_band_rows": 16}),
    "planes+bands1": dict(FORCE, **{"roi_align.band_channels": 1}),
    "planes+bands2": dict(FORCE, **{"roi_align.band_channels": 2, "roi_align.min_band_rows": 16}),
    "auto": dict(BASE, **{"roi_align.shared_staging": 1}),
    "default": dict(BASE),
}
out = {}
only = sys.argv[2].split(",") if len(sys.argv) > 2 else list(ROUTES)
for dt in (torch.float32, torch.bfloat16):
    fl = [[f.to(dt) for f in s["flist"]] for s in sets]
    for P in (7, 14):
        for name in only:
            for k, v in ROUTES[name].items():
                torch.ops.tvmi.set_option(k, v)
            args = (P, P, 2, False, 2, 5, 224.0, 4.0, 1e-6)
            with torch.no_grad():
                ms = tm(lambda i: torch.ops.tvmi.multiscale_roi_align(fl[i % 4], sets[i % 4]["rois"], scales, *args))
            out[f"{str(dt).split('.')[-1]}_{P}x{P}_{name}"] = round(ms, 4)
            print(f"{dt} {P} {name}: {ms:.4f} ms", flush=True)
for k, v in BASE.items():
    torch.ops.tvmi.set_option(k, v)
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
