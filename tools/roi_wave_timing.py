#!/usr/bin/env python3
"""Times the register-staged ("wave") RoIAlign forward kernels — the route of generic pooled shapes / adaptive sampling
(torchvision.ops.roi_align's default sampling_ratio=-1), of calls without a workspace and of the mop-up launch — on the
config-2-shaped single-level workload of the RoIPool rows (4 x 256 x 100 x 168 map, 4000 RoIs grouped by image) and on
the config-2 FPN inputs.  Per-call HIP events, median of `n`.
usage: roi_wave_timing.py out.json [label]"""
import ctypes
import json
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import vision_amd  # noqa: E402
import bench  # noqa: E402
from helpers import rois_for  # noqa: E402

dev = torch.device("cuda")
tv = torch.ops.torchvision
lib = vision_amd._loader.kernels()


def tm(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return round(statistics.median(ts), 4)


DT = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}


def abi_dtype(dt):
    # include/tvmi.h: enum tvmi_dtype
    names = {torch.float32: "TVMI_F32", torch.float16: "TVMI_F16", torch.bfloat16: "TVMI_BF16"}
    import re
    hdr = open(os.path.join(ROOT, "include", "tvmi.h")).read()
    return int(re.search(names[dt] + r"\s*=\s*(\d+)", hdr).group(1))


def no_workspace_call(x, rois, out, P, scale):
    """tvmi_roi_align_forward with workspace = NULL: the fast shapes then run on the wave kernels alone."""
    N, C, H, W = x.shape
    st = lib.tvmi_roi_align_forward(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(rois.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                                    ctypes.c_int(abi_dtype(x.dtype)), ctypes.c_int64(N), ctypes.c_int64(C), ctypes.c_int64(H), ctypes.c_int64(W),
                                    ctypes.c_int64(rois.shape[0]), ctypes.c_int64(P), ctypes.c_int64(P), ctypes.c_double(scale),
                                    ctypes.c_int64(2), ctypes.c_int(0), ctypes.c_void_p(0), ctypes.c_size_t(0),
                                    ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert st == 0, st


out = {"label": sys.argv[2] if len(sys.argv) > 2 else ""}
g = torch.Generator().manual_seed(3)
x32 = torch.randn(4, 256, 100, 168, generator=g).to(dev)
rois32 = rois_for(4, 4000, 1344, 800, 32, 400, g)
rois32 = rois32[torch.argsort(rois32[:, 0], stable=True)].to(dev)
feats, boxes, _ = bench.make_inputs(dev, 1000)
scales = [1.0 / s for s in bench.STRIDES]
from vision_amd.poolers import _convert_to_roi_format  # noqa: E402
ms_rois = _convert_to_roi_format(boxes).float()
with torch.no_grad():
    for dt in (torch.float32, torch.bfloat16, torch.float16):
        x, rois = x32.to(dt), rois32.to(dt)
        tag = str(dt).split(".")[-1]
        out[f"single_7x7_adaptive_{tag}"] = tm(lambda: tv.roi_align(x, rois, 0.125, 7, 7, -1, False))
        out[f"single_5x5_sr2_{tag}"] = tm(lambda: tv.roi_align(x, rois, 0.125, 5, 5, 2, False))
        out[f"single_7x7_sr2_dma_{tag}"] = tm(lambda: tv.roi_align(x, rois, 0.125, 7, 7, 2, False))
        for P in (7, 14):
            o = torch.empty(4000, 256, P, P, device=dev, dtype=dt)
            out[f"single_{P}x{P}_sr2_no_workspace_{tag}"] = tm(lambda: no_workspace_call(x, rois, o, P, 0.125))
        fl = [feats[str(j)].to(dt) for j in range(4)]
        out[f"fpn_7x7_adaptive_{tag}"] = tm(lambda: torch.ops.tvmi.multiscale_roi_align(fl, ms_rois, scales, 7, 7, 0, False, 2, 5, 224.0, 4.0, 1e-6))
        out[f"fpn_7x7_sr2_dma_{tag}"] = tm(lambda: torch.ops.tvmi.multiscale_roi_align(fl, ms_rois, scales, 7, 7, 2, False, 2, 5, 224.0, 4.0, 1e-6))
print(json.dumps(out, indent=1))
json.dump(out, open(sys.argv[1], "w"), indent=1)
