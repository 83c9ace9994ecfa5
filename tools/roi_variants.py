#!/usr/bin/env python3
"""Times the multi-scale RoIAlign forward of BASELINE config 2 under the launch routes of the LDS-DMA kernels
(tvmi_set_option: roi_align.pin_chunks / order / order_bands) for 7x7 / 14x14, fp32 / bf16, NCHW; plus the
channels_last kernel.  4 rotated input sets, HIP events, every route timed `reps` times interleaved (min and median kept).
usage: roi_variants.py out.json [route,route,...]"""
import json
import os
import statistics
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


KEYS = ("roi_align.pin_chunks", "roi_align.order", "roi_align.order_bands")
SAVED = {k: int(torch.ops.tvmi.get_option(k)) for k in KEYS}


def R(pin, order, bands):
    return dict(zip(KEYS, (pin, order, bands)))


ROUTES = {
    "ranges": R(0, 0, 16),
    "pinned": R(1, 0, 16),
    "pinned+order1": R(1, 1, 1),
    "pinned+order16": R(1, 1, 16),
    "pinned+order64": R(1, 1, 64),
}
out = {}
only = sys.argv[2].split(",") if len(sys.argv) > 2 else list(ROUTES)
reps = 3
for dt in (torch.float32, torch.bfloat16):
    fl = [[f.to(dt) for f in s["flist"]] for s in sets]
    for P in (7, 14):
        args = (P, P, 2, False, 2, 5, 224.0, 4.0, 1e-6)
        acc = {name: [] for name in only}
        for _ in range(reps):
            for name in only:
                for k, v in ROUTES[name].items():
                    torch.ops.tvmi.set_option(k, v)
                with torch.no_grad():
                    acc[name].append(tm(lambda i: torch.ops.tvmi.multiscale_roi_align(fl[i % 4], sets[i % 4]["rois"], scales, *args)))
        for name in only:
            key = f"{str(dt).split('.')[-1]}_{P}x{P}_{name}"
            out[key] = {"min": round(min(acc[name]), 4), "median": round(statistics.median(acc[name]), 4)}
            print(f"{key}: min {min(acc[name]):.4f} median {statistics.median(acc[name]):.4f} ms", flush=True)
    if dt == torch.float32:
        for k, v in SAVED.items():
            torch.ops.tvmi.set_option(k, v)
        flcl = [[f.contiguous(memory_format=torch.channels_last) for f in s] for s in fl]
        with torch.no_grad():
            ms = [tm(lambda i: torch.ops.tvmi.multiscale_roi_align(flcl[i % 4], sets[i % 4]["rois"], scales, 7, 7, 2, False, 2, 5, 224.0, 4.0, 1e-6)) for _ in range(reps)]
        out["float32_7x7_channels_last"] = {"min": round(min(ms), 4), "median": round(statistics.median(ms), 4)}
        print("float32_7x7_channels_last", out["float32_7x7_channels_last"], flush=True)
        del flcl
for k, v in SAVED.items():
    torch.ops.tvmi.set_option(k, v)
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
