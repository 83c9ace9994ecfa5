#!/usr/bin/env python3
"""oracle/gen_golden.py — TEST INFRASTRUCTURE ONLY.

Generates the golden input/output vectors committed under tests/golden/ by running the REAL
reference CPU kernels (oracle/_ref, built from /root/reference by oracle/build_ref.py) on
seeded inputs — and, for resize, installed-torch (2.10.0) CPU `F.interpolate`, which is where
the reference's resize arithmetic lives.  Runs only where /root/reference exists; the
vectors travel with the repository.

Input constructions follow the reference's own tests:
  * RoI ops: fixed-size feature map + the 4 hand-written RoIs of test/test_ops.py:142-163
    (plus random / out-of-range RoIs),
  * NMS: the adversarial "one pair just over the threshold" generator of
    test/test_ops.py:899-914, with duplicate scores added to pin the stable tie order,
  * deform_conv2d: the asymmetric configuration of test/test_ops.py:1113-1167.
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import build_ref  # noqa: E402
from oracle import oracle as O  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: (v.numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in arrays.items()})
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


def nms_case(n, thr, seed, dup=False, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    boxes = torch.rand(n, 4, generator=g, dtype=dtype) * 100
    boxes[:, 2:] += boxes[:, :2]
    boxes[-1, :] = boxes[0, :]
    x0, y0, x1, y1 = boxes[-1].tolist()
    t = thr + 1e-5
    boxes[-1, 2] += (x1 - x0) * (1 - t) / t
    scores = torch.rand(n, generator=g, dtype=dtype)
    if dup:
        scores = (scores * 16).floor() / 16  # many exact ties
    return boxes, scores


def main():
    if not build_ref.have_reference():
        raise SystemExit("gen_golden.py needs /root/reference (it runs the reference's own kernels)")
    build_ref.build()
    assert O.load_reference()
    tv = torch.ops.torchvision

    # ---------------------------------------------------------------- NMS
    cases = {}
    i = 0
    for thr in (0.2, 0.5, 0.8):
        for seed in range(3):
            for dup in (False, True):
                b, s = nms_case(400, thr, seed, dup)
                cases[f"boxes{i}"], cases[f"scores{i}"], cases[f"thr{i}"] = b, s, np.float64(thr)
                cases[f"keep{i}"] = tv.nms(b, s, thr)
                i += 1
    b, s = nms_case(300, 0.5, 7, dtype=torch.float64)
    cases[f"boxes{i}"], cases[f"scores{i}"], cases[f"thr{i}"], cases[f"keep{i}"] = b, s, np.float64(0.5), tv.nms(b, s, 0.5)
    i += 1
    # degenerate boxes (zero area -> 0/0), negative threshold, threshold 0
    g = torch.Generator().manual_seed(11)
    b = torch.rand(200, 4, generator=g) * 50
    b[:, 2:] += b[:, :2]
    b[::7, 2:] = b[::7, :2]
    s = torch.rand(200, generator=g)
    for thr in (0.0, -0.1, 0.3, 1.0):
        cases[f"boxes{i}"], cases[f"scores{i}"], cases[f"thr{i}"], cases[f"keep{i}"] = b, s, np.float64(thr), tv.nms(b, s, thr)
        i += 1
    cases["count"] = np.int64(i)
    save("nms", **cases)

    # ---------------------------------------------------------------- RoI family
    g = torch.Generator().manual_seed(3)
    pool = 5
    x = torch.rand(2, 2 * pool * pool, 10, 10, generator=g)
    fixed = torch.tensor([[0, 0, 0, 9, 9], [0, 0, 5, 4, 9], [0, 5, 5, 9, 9], [1, 0, 0, 9, 9]], dtype=torch.float32)
    rnd = torch.cat([torch.randint(0, 2, (12, 1), generator=g).float(), torch.rand(12, 2, generator=g) * 8 - 2,
                     3 + torch.rand(12, 2, generator=g) * 10], 1)
    rois = torch.cat([fixed, rnd, torch.tensor([[1, 20, 20, 30, 30], [0, 4, 4, 4, 4], [0, 6, 7, 2, 3]], dtype=torch.float32)])
    d = {"x": x, "rois": rois}
    for scale in (1.0, 0.5):
        for sr in (-1, 2):
            for aligned in (False, True):
                key = f"s{scale}_sr{sr}_a{int(aligned)}"
                y = tv.roi_align(x, rois, scale, pool, pool, sr, aligned)
                d["roi_align_" + key] = y
                gr = torch.linspace(-1, 1, y.numel()).reshape(y.shape)
                d["roi_align_bwd_" + key] = tv._roi_align_backward(gr, rois, scale, pool, pool, 2, x.shape[1], 10, 10, sr, aligned)
            y, m = tv.ps_roi_align(x, rois[:-1], scale, pool, pool, sr)
            d[f"ps_roi_align_s{scale}_sr{sr}"], d[f"ps_roi_align_map_s{scale}_sr{sr}"] = y, m
        y, a = tv.roi_pool(x, rois, scale, pool, pool)
        d[f"roi_pool_s{scale}"], d[f"roi_pool_argmax_s{scale}"] = y, a
        y, m = tv.ps_roi_pool(x, rois, scale, pool, pool)
        d[f"ps_roi_pool_s{scale}"], d[f"ps_roi_pool_map_s{scale}"] = y, m
    save("roi_ops", **d)

    # ---------------------------------------------------------------- deform_conv2d
    g = torch.Generator().manual_seed(5)
    B, C, OC, H, W, kh, kw = 4, 6, 2, 5, 4, 3, 2
    stride, pad, dil, groups, og = (2, 1), (1, 0), (2, 1), 2, 3
    oh = (H + 2 * pad[0] - (dil[0] * (kh - 1) + 1)) // stride[0] + 1
    ow = (W + 2 * pad[1] - (dil[1] * (kw - 1) + 1)) // stride[1] + 1
    x = torch.rand(B, C, H, W, generator=g)
    w = torch.randn(OC, C // groups, kh, kw, generator=g)
    off = torch.randn(B, og * 2 * kh * kw, oh, ow, generator=g)
    m = torch.randn(B, og * kh * kw, oh, ow, generator=g)
    bias = torch.randn(OC, generator=g)
    args = (stride[0], stride[1], pad[0], pad[1], dil[0], dil[1], groups, og)
    d = {"x": x, "weight": w, "offset": off, "mask": m, "bias": bias,
         "out_mask": tv.deform_conv2d(x, w, off, m, bias, *args, True),
         "out_nomask": tv.deform_conv2d(x, w, off, torch.zeros(B, 1), bias, *args, False)}
    gr = torch.linspace(-1, 1, d["out_mask"].numel()).reshape(d["out_mask"].shape)
    for nm, t in zip(("gin", "gw", "goff", "gmask", "gbias"), tv._deform_conv2d_backward(gr, x, w, off, m, bias, *args, True)):
        d["bwd_" + nm] = t
    save("deform_conv2d", **d)

    # ---------------------------------------------------------------- rotated IoU
    g = torch.Generator().manual_seed(9)
    b1 = torch.cat([torch.rand(40, 2, generator=g) * 60, 4 + torch.rand(40, 2, generator=g) * 30,
                    torch.rand(40, 1, generator=g) * 360 - 180], 1)
    b2 = torch.cat([torch.rand(33, 2, generator=g) * 60, 4 + torch.rand(33, 2, generator=g) * 30,
                    torch.rand(33, 1, generator=g) * 360 - 180], 1)
    b2[:6] = b1[:6]
    b2[6, 2] = 0.0
    save("box_iou_rotated", b1=b1, b2=b2, iou=tv.box_iou_rotated(b1, b2),
         iou64=tv.box_iou_rotated(b1.double(), b2.double()))

    # ---------------------------------------------------------------- resize (torch 2.10 CPU)
    g = torch.Generator().manual_seed(13)
    img = torch.rand(2, 3, 37, 53, generator=g)
    d = {"img": img, "torch_version": np.bytes_(torch.__version__.encode())}
    for mode in ("nearest", "nearest-exact", "bilinear", "bicubic"):
        for size in ((19, 31), (80, 61), (37, 53)):
            for aa in ((False, True) if mode in ("bilinear", "bicubic") else (False,)):
                ac = None if mode.startswith("nearest") else False
                d[f"{mode}_{size[0]}x{size[1]}_aa{int(aa)}"] = F.interpolate(img, size=size, mode=mode, align_corners=ac, antialias=aa)
        if not mode.startswith("nearest"):
            d[f"{mode}_80x61_ac1"] = F.interpolate(img, size=(80, 61), mode=mode, align_corners=True)
    save("resize", **d)


if __name__ == "__main__":
    main()
