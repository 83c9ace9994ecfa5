"""Shared input generators for the test-suite (SURVEY.md §8d constructions)."""
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def gen(seed=0):
    return torch.Generator().manual_seed(seed)


def random_boxes(n, width, height, lo, hi, g, dtype=torch.float32):
    xy = torch.rand(n, 2, generator=g) * torch.tensor([max(width - 64.0, 1.0), max(height - 64.0, 1.0)])
    wh = lo + torch.rand(n, 2, generator=g) * (hi - lo)
    x2y2 = torch.minimum(xy + wh, torch.tensor([float(width), float(height)]))
    return torch.cat([xy, x2y2], 1).to(dtype)


def adversarial_nms_inputs(n, thr, g, dup=False):
    """One pair sits just over the IoU threshold (reference test/test_ops.py:899-914)."""
    boxes = torch.rand(n, 4, generator=g) * 100
    boxes[:, 2:] += boxes[:, :2]
    boxes[-1, :] = boxes[0, :]
    x0, y0, x1, y1 = boxes[-1].tolist()
    t = thr + 1e-5
    boxes[-1, 2] += (x1 - x0) * (1 - t) / t
    scores = torch.rand(n, generator=g)
    if dup:
        scores = (scores * 16).floor() / 16
    return boxes, scores


def rois_for(n_img, k, width, height, lo, hi, g, dtype=torch.float32):
    b = random_boxes(k, width, height, lo, hi, g)
    idx = torch.randint(0, n_img, (k, 1), generator=g).float()
    return torch.cat([idx, b], 1).to(dtype)


def fpn_features(batch, channels, height, width, g, dtype=torch.float32, levels=(4, 8, 16, 32)):
    feats = {}
    for i, s in enumerate(levels):
        feats[str(i)] = torch.randn(batch, channels, -(-height // s), -(-width // s), generator=g).to(dtype)
    return feats


def python_greedy_nms(boxes, scores, thr):
    """Independent greedy loop on torch ops (what the reference's TestNMS._reference_nms does)."""
    from vision_amd.boxes import box_iou

    picked = []
    _, idx = scores.sort(descending=True, stable=True)
    while idx.numel() > 0:
        cur = idx[0]
        picked.append(cur.item())
        if idx.numel() == 1:
            break
        idx = idx[1:]
        iou = box_iou(boxes[idx], boxes[cur].unsqueeze(0)).squeeze(1)
        idx = idx[iou <= thr]
    return torch.as_tensor(picked, dtype=torch.int64)
