#!/usr/bin/env python3
"""Host-side overlap factor of the ordered RoI sequence of bench.py's config 2 ("first measure, host-side and
for free"): for groups of G order-adjacent RoIs of one (image, level) — the order of roi_fwd_order: image, level, window-top band —
    factor(G) = sum over the RoIs of a group of (window rows x window columns)  /  area of the union of those windows
i.e. how many times a feature-map pixel of the group's union is staged when every RoI stages its own window (what the DMA kernel
does per channel), against staging ONE union window per group.  Also per level: share of the RoIs, of the window pixels (= of the
line accesses, roughly) and the factor when a whole (image, level) plane is staged once."""
import math
import sys

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench  # noqa: E402


def windows(boxes, scale, H, W, P=7, sr=2):
    """(y0, y1, x0, x1) inclusive pixel bounds of the bilinear taps of a RoI (roi_align_common.h:32-124, aligned=False)."""
    x1, y1, x2, y2 = [boxes[:, i].double() * scale for i in range(4)]
    w, h = (x2 - x1).clamp(min=1.0), (y2 - y1).clamp(min=1.0)
    bw, bh = w / P, h / P
    ys0 = y1 + 0.5 * bh / sr
    ys1 = y1 + (P - 1) * bh + (sr - 0.5) * bh / sr
    xs0 = x1 + 0.5 * bw / sr
    xs1 = x1 + (P - 1) * bw + (sr - 0.5) * bw / sr
    y0 = ys0.floor().clamp(0, H - 1).long()
    yb = (ys1.floor() + 1).clamp(0, H - 1).long()
    x0 = xs0.floor().clamp(0, W - 1).long()
    xb = (xs1.floor() + 1).clamp(0, W - 1).long()
    return torch.stack([y0, yb, x0, xb], 1)


def main():
    feats, boxes, _ = bench.make_inputs("cpu", 1000)
    strides = bench.STRIDES
    tot = {G: [0.0, 0.0] for G in (1, 2, 4, 8, 16)}
    per_level = {}
    for img, b in enumerate(boxes):
        area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
        lvl = torch.floor(4 + torch.log2(torch.sqrt(area) / 224) + 1e-6).clamp(2, 5).long() - 2
        for l, s in enumerate(strides):
            sel = (lvl == l).nonzero()[:, 0]
            if sel.numel() == 0:
                continue
            H, W = bench.IMG_H // s, bench.IMG_W // s
            win = windows(b[sel], 1.0 / s, H, W)
            band = (win[:, 0].float() * 64 / H).long()
            order = torch.argsort(band, stable=True)        # the pre-pass: 64 window-top bands per (image, level)
            win = win[order]
            px = ((win[:, 1] - win[:, 0] + 1) * (win[:, 3] - win[:, 2] + 1)).double()
            plane = np.zeros((H, W), dtype=bool)
            for y0, yb, x0, xb in win.tolist():
                plane[y0:yb + 1, x0:xb + 1] = True
            d = per_level.setdefault(l, dict(rois=0, px=0.0, union=0.0, plane=0.0, rows=0.0))
            d["rois"] += sel.numel()
            d["px"] += float(px.sum())
            d["union"] += float(plane.sum())
            d["plane"] += H * W
            d["rows"] += float((win[:, 1] - win[:, 0] + 1).sum())
            for G in tot:
                for g0 in range(0, win.shape[0], G):
                    grp = win[g0:g0 + G]
                    y0, yb, x0, xb = int(grp[:, 0].min()), int(grp[:, 1].max()), int(grp[:, 2].min()), int(grp[:, 3].max())
                    m = np.zeros((yb - y0 + 1, xb - x0 + 1), dtype=bool)
                    for a, bb, c, dd in grp.tolist():
                        m[a - y0:bb - y0 + 1, c - x0:dd - x0 + 1] = True
                    tot[G][0] += float(px[g0:g0 + G].sum())
                    tot[G][1] += float(m.sum())
    allpx = sum(d["px"] for d in per_level.values())
    print("level  rois   share_of_window_px  mean_window_px  px/union(whole level)  px/plane")
    for l, d in sorted(per_level.items()):
        print(f"P{l + 2}   {d['rois']:5d}   {d['px'] / allpx:6.3f}            {d['px'] / d['rois']:8.1f}        {d['px'] / d['union']:6.2f}               {d['px'] / d['plane']:6.2f}")
    print("groups of G order-adjacent RoIs (same image and level): sum of window px / union px")
    for G, (a, b) in tot.items():
        print(f"  G = {G:2d}: {a / b:5.2f}")


if __name__ == "__main__":
    main()
