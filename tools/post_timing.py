#!/usr/bin/env python3
"""tools/post_timing.py — wall time per call of the fused detector post-processing at Mask R-CNN sizes (2 images):
filter_proposals, postprocess_detections (thresholds 0.0 / 0.05; list and padded forms) and their host-side pieces."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vision_amd  # noqa: E402
from vision_amd import detection_post as dp  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)


def rb(n, w, h):
    xy = torch.rand(n, 2, generator=g) * torch.tensor([w - 40.0, h - 40.0])
    wh = 8 + torch.rand(n, 2, generator=g) * 300
    return torch.cat([xy, torch.minimum(xy + wh, torch.tensor([float(w), float(h)]))], 1)


def wall(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return round((time.perf_counter() - t0) / n * 1e3, 3)


B = 2
shapes = [(800, 1333)] * B
props = [rb(1000, 1333, 800).to(dev) for _ in range(B)]
logits = (torch.randn(B * 1000, 91, generator=g) * 0.1).to(dev)
reg = (torch.randn(B * 1000, 364, generator=g) * 0.5).to(dev)
A_lvls = [200 * 336 * 3, 100 * 168 * 3, 50 * 84 * 3, 25 * 42 * 3, 13 * 21 * 3]
A = sum(A_lvls)
anchors = rb(B * A, 1333, 800).reshape(B, A, 4).to(dev)
obj = torch.randn(B, A, generator=g).to(dev)
out = {}
for thr in (0.0, 0.05):
    out[f"postprocess_detections_thr{thr}_list_ms"] = wall(lambda: vision_amd.postprocess_detections(logits, reg, props, shapes, score_thresh=thr))
    out[f"postprocess_detections_thr{thr}_padded_ms"] = wall(lambda: vision_amd.postprocess_detections(logits, reg, props, shapes, score_thresh=thr, padded=True))
out["filter_proposals_list_ms"] = wall(lambda: vision_amd.filter_proposals(anchors, obj, shapes, A_lvls, pre_nms_top_n=1000, post_nms_top_n=1000))
out["filter_proposals_padded_ms"] = wall(lambda: vision_amd.filter_proposals(anchors, obj, shapes, A_lvls, pre_nms_top_n=1000, post_nms_top_n=1000, padded=True))
out["host_to_device_small_ms"] = wall(lambda: dp._image_hw(shapes, dev))
out["pageable_tensor_to_device_ms"] = wall(lambda: torch.tensor([[800.0, 1333.0]] * B, device=dev))
n = B * 1000 * 90
b, s = torch.rand(n, 4, device=dev), torch.rand(n, device=dev)
b[:, 2:] += b[:, :2]
seg = torch.randint(0, B * 91, (n,), device=dev)
for frac in (1.0, 0.03):
    valid = (torch.rand(n, device=dev) < frac).to(torch.uint8)
    out[f"nms_segmented_masked_180k_live{frac}_ms"] = wall(lambda: torch.ops.tvmi.nms_segmented_masked(b, s, seg, valid, 0.5, B * 91))
    sel = valid.nonzero()[:, 0]
    bs, ss, sg = b[sel], s[sel], seg[sel]
    out[f"nms_segmented_compacted_live{frac}_ms"] = wall(lambda: torch.ops.tvmi.nms_segmented(bs, ss, sg, 0.5, B * 91))
print(json.dumps(out, indent=1))
