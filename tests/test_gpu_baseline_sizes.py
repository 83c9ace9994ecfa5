"""Parity at BASELINE.json's stated sizes (`-m gpu`, MI355X) — the HIP kernels against the reference's own CPU
kernels (oracle/_ref, when the prebuilt library travelled) or the C restatement (oracle/tvmi_oracle.c), on the
exact configurations the metric is quoted on:

  config 2  MultiScaleRoIAlign, 4 images x 4 FPN levels x 256 channels, 4 x 1000 proposals, 7x7 and 14x14,
            fp32 and bf16, forward and the fused backward            (reference test: test/test_ops.py:287-318)
  config 3  nms / batched_nms, 100,000 boxes, 80 classes, IoU 0.5, sparse canvas and the dense 200x200 variant,
            index lists bit-exact                                     (test/test_ops.py:916-925, 1026-1044)
  config 4  deform_conv2d 2x256x100x136, k3, groups 1 / 256, with and without mask (test/test_ops.py:1287-1319)
  RoIPool   the shape DESIGN.md quotes its timing on (4x256x100x168, 4000 RoIs, 7x7): value and argmax bit-exact

Bars: NMS index lists identical; fp32 values within 1e-4 (BASELINE.json north_star); bf16 within the reference's
own 5e-3 (test/test_ops.py:139-140) against the fp32 result on the rounded inputs; backward sums within
1e-4 * sqrt(contributors) * scale (summation order of ~10^2 contributions per pixel differs from the CPU loop).
The CPU side costs about a minute in total."""
import math

import numpy as np
import pytest
import torch

import vision_amd
from oracle import oracle as O
from helpers import gen, random_boxes

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 1e-4

IMG_H, IMG_W, CH, BATCH, PROPS = 800, 1344, 256, 4, 1000
STRIDES = (4, 8, 16, 32)


def _cpu_nms(boxes, scores, thr, idxs=None):
    """Reference CPU kernel when oracle/_ref is loaded (single class), else / for segments the C restatement."""
    if idxs is None and O.load_reference():
        return torch.ops.torchvision.nms(boxes, scores, thr).numpy()
    return O.nms(boxes.numpy(), scores.numpy(), thr, None if idxs is None else idxs.numpy())


def _config3(canvas, seed):
    g = gen(seed)
    n = 100_000
    boxes = random_boxes(n, canvas, canvas, 1, 101, g)
    scores = torch.rand(n, generator=g)
    idxs = torch.randint(0, 80, (n,), generator=g)
    return boxes, scores, idxs


@pytest.mark.parametrize("canvas", [1000, 200], ids=["sparse", "dense"])
def test_config3_nms_100k_bit_exact(tv, canvas):
    boxes, scores, _ = _config3(canvas, 300 + canvas)
    keep = tv.nms(boxes.to(DEV), scores.to(DEV), 0.5).cpu().numpy()
    want = _cpu_nms(boxes, scores, 0.5)
    assert keep.shape == want.shape and np.array_equal(keep, want)
    if canvas == 200:
        assert len(want) < 20_000      # the dense variant really suppresses most boxes


@pytest.mark.parametrize("canvas", [1000, 200], ids=["sparse", "dense"])
def test_config3_batched_nms_100k_x80_bit_exact(canvas):
    boxes, scores, idxs = _config3(canvas, 400 + canvas)
    keep = vision_amd.batched_nms(boxes.to(DEV), scores.to(DEV), idxs.to(DEV), 0.5).cpu().numpy()
    want = _cpu_nms(boxes, scores, 0.5, idxs)
    assert keep.shape == want.shape and np.array_equal(keep, want)


@pytest.mark.parametrize("groups", [1, 256])
@pytest.mark.parametrize("use_mask", [False, True])
def test_config4_deform_conv2d_full_size(tv, groups, use_mask):
    g = gen(500 + groups)
    B, C, H, W, OC = 2, 256, 100, 136, 256
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(OC, C // groups, 3, 3, generator=g) * 0.01
    off = torch.randn(B, 18, H, W, generator=g)
    m = torch.rand(B, 9, H, W, generator=g)
    b = torch.randn(OC, generator=g)
    y = vision_amd.deform_conv2d(x.to(DEV), off.to(DEV), w.to(DEV), b.to(DEV), (1, 1), (1, 1), (1, 1),
                                 m.to(DEV) if use_mask else None).cpu().numpy()
    if O.load_reference():
        ref = torch.ops.torchvision.deform_conv2d(x, w, off, m if use_mask else torch.zeros(B, 1), b, 1, 1, 1, 1, 1, 1,
                                                  groups, 1, use_mask).numpy()
    else:
        ref = O.deform_conv2d(x.numpy(), w.numpy(), off.numpy(), m.numpy(), b.numpy(), (1, 1), (1, 1), (1, 1), groups, 1, use_mask)
    np.testing.assert_allclose(y, ref, rtol=0, atol=TOL)


def _config2(seed=1000):
    g = gen(seed)
    feats = [torch.randn(BATCH, CH, IMG_H // s, IMG_W // s, generator=g) for s in STRIDES]
    boxes = []
    for _ in range(BATCH):  # the proposal generator of bench.py (all four levels receive boxes)
        xy = torch.rand(PROPS, 2, generator=g) * torch.tensor([IMG_W - 64.0, IMG_H - 64.0])
        side = torch.exp(torch.rand(PROPS, generator=g) * (math.log(640.0) - math.log(16.0)) + math.log(16.0))
        aspect = torch.exp((torch.rand(PROPS, generator=g) * 2 - 1) * math.log(3.0))
        wh = torch.stack([side * aspect.sqrt(), side / aspect.sqrt()], 1)
        boxes.append(torch.cat([xy, torch.minimum(xy + wh, torch.tensor([float(IMG_W), float(IMG_H)]))], 1))
    return feats, boxes


def _cpu_roi_align(x, rois, scale, P):
    if O.load_reference():
        return torch.ops.torchvision.roi_align(x, rois, scale, P, P, 2, False).numpy()
    return O.roi_align(x.numpy(), rois.numpy(), scale, P, P, 2, False)


def _cpu_roi_align_backward(gr, rois, scale, P, shape):
    N, C, H, W = shape
    if O.load_reference():
        return torch.ops.torchvision._roi_align_backward(gr, rois, scale, P, P, N, C, H, W, 2, False).numpy()
    return O.roi_align_backward(gr.numpy(), rois.numpy(), scale, P, P, N, C, H, W, 2, False)


def test_config1_full_shape_roi_align_and_nms(tv):
    """BASELINE config 1 at its FULL shape on the GPU (VERDICT r05 missing 6; the CPU form is the reference's own plumbing case):
    ONE 1x256x200x272 fp32 map, 1000 random boxes / scores, roi_align 7x7 (scale 1/4, sampling_ratio 2) within 1e-4 of the
    reference CPU kernel, nms(1000) index list identical — the inputs bench.py's `configs.config1_*` rows are timed on."""
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 256, 200, 272, generator=g)
    xy = torch.rand(1000, 2, generator=g) * torch.tensor([1088 - 64.0, 800 - 64.0])
    wh = 16 + torch.rand(1000, 2, generator=g) * 284
    rois = torch.cat([torch.zeros(1000, 1), xy, torch.minimum(xy + wh, torch.tensor([1088.0, 800.0]))], 1)
    scores = torch.rand(1000, generator=g)
    got = tv.roi_align(x.to(DEV), rois.to(DEV), 0.25, 7, 7, 2, False).cpu().numpy()
    np.testing.assert_allclose(got, _cpu_roi_align(x, rois, 0.25, 7), rtol=0, atol=TOL)
    boxes = rois[:, 1:].contiguous()
    keep = tv.nms(boxes.to(DEV), scores.to(DEV), 0.5).cpu().numpy()
    want = _cpu_nms(boxes, scores, 0.5)
    assert np.array_equal(keep, want) and 100 < len(want) < 1000


@pytest.mark.parametrize("P", [7, 14])
def test_config2_multiscale_roi_align_full_fwd_bwd(P):
    feats, boxes = _config2()
    names = [str(i) for i in range(4)]
    pool = vision_amd.MultiScaleRoIAlign(names, P, 2)
    dfeats = {n: f.to(DEV).requires_grad_(True) for n, f in zip(names, feats)}
    dboxes = [b.to(DEV) for b in boxes]
    out = pool(dfeats, dboxes, [(IMG_H, IMG_W)] * BATCH)
    assert tuple(out.shape) == (BATCH * PROPS, CH, P, P)
    levels = pool.map_levels(boxes)
    rois = torch.cat([torch.cat([torch.full((PROPS, 1), float(i)), b], 1) for i, b in enumerate(boxes)])
    g = gen(7)
    grad = torch.randn(out.shape, generator=g)
    out.backward(grad.to(DEV))
    got = out.detach().cpu().numpy()
    for lvl in range(4):
        sel = torch.nonzero(levels == lvl)[:, 0]
        assert sel.numel() > 0
        ref = _cpu_roi_align(feats[lvl], rois[sel], pool.scales[lvl], P)
        np.testing.assert_allclose(got[sel.numpy()], ref, rtol=0, atol=TOL, err_msg=f"forward level {lvl}")
        refb = _cpu_roi_align_backward(grad[sel].contiguous(), rois[sel], pool.scales[lvl], P, tuple(feats[lvl].shape))
        gb = dfeats[names[lvl]].grad.cpu().numpy()
        # a pixel sums up to a few hundred contributions of magnitude <~ 4: order-of-summation noise only
        np.testing.assert_allclose(gb, refb, rtol=1e-4, atol=TOL * max(1.0, float(np.abs(refb).max())),
                                   err_msg=f"backward level {lvl}")
    # bf16 maps through the same op: bar = the reference's bf16 tolerance against fp32 on the rounded inputs
    bfeats = {n: f.to(torch.bfloat16).to(DEV) for n, f in zip(names, feats)}
    with torch.no_grad():
        ob = pool(bfeats, dboxes, [(IMG_H, IMG_W)] * BATCH)
    assert ob.dtype == torch.bfloat16
    rb = rois.clone()
    for lvl in range(4):
        sel = torch.nonzero(levels == lvl)[:, 0]           # all 4000 RoIs, like the fp32 path
        ref = _cpu_roi_align(feats[lvl].to(torch.bfloat16).float(), rb[sel], pool.scales[lvl], P)
        np.testing.assert_allclose(ob[sel.to(DEV)].float().cpu().numpy(), ref, rtol=5e-3, atol=5e-3, err_msg=f"bf16 level {lvl}")


def test_config2_schema_ops_match_fused_op(tv):
    """What the unchanged reference python computes on this library (per-level torchvision::roi_align + index_put,
    poolers.py:199-222) equals the fused single-launch op bit for bit."""
    feats, boxes = _config2(seed=1001)
    feats = [f[:, :64].contiguous() for f in feats]
    names = [str(i) for i in range(4)]
    pool = vision_amd.MultiScaleRoIAlign(names, 7, 2)
    with torch.no_grad():
        fused = pool({n: f.to(DEV) for n, f in zip(names, feats)}, [b.to(DEV) for b in boxes], [(IMG_H, IMG_W)] * BATCH)
    levels = pool.map_levels(boxes).to(DEV)
    rois = torch.cat([torch.cat([torch.full((PROPS, 1), float(i)), b], 1) for i, b in enumerate(boxes)]).to(DEV)
    result = torch.zeros_like(fused)
    for lvl in range(4):
        idx = torch.where(levels == lvl)[0]
        result[idx] = tv.roi_align(feats[lvl].to(DEV), rois[idx], pool.scales[lvl], 7, 7, 2, False)
    assert torch.equal(result, fused)


def test_roi_pool_measured_shape_bit_exact(tv):
    """RoIPool 7x7 at the shape its timing is quoted on (tools/gpu_matrix.py `roipool`): 4x256x100x168 fp32, 4000 RoIs of
    32-400 px at scale 1/8 grouped by image — pooled values and argmax identical to the reference CPU kernel
    (cpu/roi_pool_kernel.cpp), incl. ties (values rounded to quarter steps)."""
    from helpers import rois_for

    g = gen(3)
    x = (torch.randn(4, 256, 100, 168, generator=g) * 4).round() / 4
    rois = rois_for(4, 4000, 1344, 800, 32, 400, g)
    rois = rois[torch.argsort(rois[:, 0], stable=True)]
    y, a = tv.roi_pool(x.to(DEV), rois.to(DEV), 0.125, 7, 7)
    if O.load_reference():
        ry, ra = torch.ops.torchvision.roi_pool(x, rois, 0.125, 7, 7)
        ry, ra = ry.numpy(), ra.numpy()
    else:
        ry, ra = O.roi_pool(x.numpy(), rois.numpy(), 0.125, 7, 7)
    assert np.array_equal(a.cpu().numpy(), ra)
    assert np.array_equal(y.cpu().numpy(), ry)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def test_config2_roi_align_backward_16bit_values(tv, dtype):
    """Config 2 says "fwd/bwd ... fp32 vs bf16": the VALUES of the 16-bit backward, fused (one multi-scale launch) and per
    level (`torchvision::_roi_align_backward`), against the reference CPU kernel run in fp32 on the same rounded
    gradients.  Ours accumulates in fp32 and rounds each pixel once (the reference's CUDA kernel adds 16-bit atomics,
    cuda/roi_align_kernel.cu:304-327); bar = the reference's own 16-bit tolerance 5e-3 (test/test_ops.py:139-140)."""
    P = 7
    feats, boxes = _config2(seed=1002)
    names = [str(i) for i in range(4)]
    pool = vision_amd.MultiScaleRoIAlign(names, P, 2)
    dfeats = {n: f.to(dtype).to(DEV).requires_grad_(True) for n, f in zip(names, feats)}
    dboxes = [b.to(DEV) for b in boxes]
    out = pool(dfeats, dboxes, [(IMG_H, IMG_W)] * BATCH)
    assert out.dtype == dtype
    grad = torch.randn(out.shape, generator=gen(8)).to(dtype)
    out.backward(grad.to(DEV))
    levels = pool.map_levels(boxes)
    rois = torch.cat([torch.cat([torch.full((PROPS, 1), float(i)), b], 1) for i, b in enumerate(boxes)])
    for lvl in range(4):
        sel = torch.nonzero(levels == lvl)[:, 0]
        shape = tuple(feats[lvl].shape)
        refb = _cpu_roi_align_backward(grad[sel].float().contiguous(), rois[sel], pool.scales[lvl], P, shape)
        bar = 5e-3 * max(1.0, float(np.abs(refb).max()))
        gb = dfeats[names[lvl]].grad
        assert gb.dtype == dtype
        np.testing.assert_allclose(gb.float().cpu().numpy(), refb, rtol=5e-3, atol=bar, err_msg=f"fused backward level {lvl}")
        # the schema op the reference's autograd formula calls (_autograd_registrations.py:30-60), 16-bit rois like the reference passes
        r16 = rois[sel].to(dtype)
        got = tv._roi_align_backward(grad[sel].to(DEV), r16.to(DEV), pool.scales[lvl], P, P, *shape, 2, False)
        assert got.dtype == dtype
        ref16 = _cpu_roi_align_backward(grad[sel].float().contiguous(), r16.float(), pool.scales[lvl], P, shape)
        np.testing.assert_allclose(got.float().cpu().numpy(), ref16, rtol=5e-3, atol=5e-3 * max(1.0, float(np.abs(ref16).max())),
                                   err_msg=f"per-level backward level {lvl}")


# bf16 `_deform_conv2d_backward` at config 4: worst |error| / max |reference| per gradient measured on the MI355X, + 20 %
BF16_BWD_BAR = {  # measured (round 6): g=1 3.14e-3 / 3.69e-3 / 2.27e-3 / 3.40e-3 / 2.17e-3; g=256 2.71e-3 / 2.88e-3 / 2.45e-3 / 2.44e-3 / 3.63e-3
    1: {"grad_input": 3.8e-3, "grad_weight": 4.5e-3, "grad_offset": 2.8e-3, "grad_mask": 4.1e-3, "grad_bias": 2.7e-3},
    256: {"grad_input": 3.3e-3, "grad_weight": 3.5e-3, "grad_offset": 3.0e-3, "grad_mask": 3.0e-3, "grad_bias": 4.4e-3}}


@pytest.mark.parametrize("groups", [1, 256])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
def test_config4_deform_conv2d_backward_full_size(tv, groups, dtype):
    """`_deform_conv2d_backward` at config-4 size (2x256x100x136, k3, 256 -> 256, mask on) against the
    reference CPU kernel (cpu/deform_conv2d_kernel.cpp:1153-1226) — all five gradients.  fp32: sums of up to 2304 (g=1) /
    27,200 (weights: over the pixels) products in another order, bar 1e-4 relative to the gradient's own scale.  bf16: the
    reference computes the 16-bit backward in the 16-bit type; ours is compared with the fp32 reference on the rounded
    tensors: the `columns` intermediate is a 16-bit tensor (in the reference too, cuda/deform_conv2d_kernel.cu:752-1033) and
    the result is rounded once more: the bar is absolute, per gradient, = the worst error measured + 20 % (BF16_BWD_BAR; round 5
    had one 3e-2 bar with an rtol of the same size on top)."""
    g = gen(620 + groups)
    B, C, H, W, OC = 2, 256, 100, 136, 256
    x = torch.randn(B, C, H, W, generator=g).to(dtype)
    w = (torch.randn(OC, C // groups, 3, 3, generator=g) * (0.01 if groups == 1 else 0.2)).to(dtype)
    off = torch.randn(B, 18, H, W, generator=g).to(dtype)
    m = torch.rand(B, 9, H, W, generator=g).to(dtype)
    b = torch.randn(OC, generator=g).to(dtype)
    gr = torch.randn(B, OC, H, W, generator=g).to(dtype)
    if not O.load_reference():
        pytest.skip("needs the reference CPU kernels (oracle/_ref): the C restatement has no deform_conv2d backward")
    got = tv._deform_conv2d_backward(gr.to(DEV), x.to(DEV), w.to(DEV), off.to(DEV), m.to(DEV), b.to(DEV), 1, 1, 1, 1, 1, 1, groups, 1, True)
    if True:
        ref = torch.ops.torchvision._deform_conv2d_backward(gr.float(), x.float(), w.float(), off.float(), m.float(), b.float(),
                                                            1, 1, 1, 1, 1, 1, groups, 1, True)
        ref = [r.numpy() for r in ref]
    measured = {}
    for name, a, r in zip(("grad_input", "grad_weight", "grad_offset", "grad_mask", "grad_bias"), got, ref):
        assert a.dtype == dtype, name
        scale = max(1.0, float(np.abs(r).max()))
        err = float(np.abs(a.float().cpu().numpy() - r).max()) / scale
        measured[name] = err
        if dtype == torch.float32:
            np.testing.assert_allclose(a.float().cpu().numpy(), r, rtol=1e-4, atol=1e-4 * scale, err_msg=f"{name} groups={groups}")
        else:
            # an absolute bar only (no rtol on top), per gradient = the worst error measured on the MI355X
            # (round 6, printed below with -s) + 20 %, in units of the gradient's largest magnitude
            assert err <= BF16_BWD_BAR[groups][name], (name, groups, err, BF16_BWD_BAR[groups][name])
    print(f"deform_conv2d backward {dtype} groups={groups}: max |err| / scale per gradient", {k: f"{v:.2e}" for k, v in measured.items()})


@pytest.mark.parametrize("dtype,tol", [(torch.bfloat16, 1e-2), (torch.float16, 2e-3)], ids=["bf16", "fp16"])
@pytest.mark.parametrize("groups", [1, 256])
def test_config4_deform_conv2d_16bit(dtype, tol, groups):
    """16-bit deform_conv2d at config 4 (timed in profiles/*matrix.json) against the reference CPU kernel in fp32 on the
    rounded tensors.  The reference keeps `columns` and the addmm_ accumulation in the 16-bit type
    (cuda/deform_conv2d_kernel.cu:1234-1239); ours accumulates in fp32 (exact 16-bit products on the MFMA path) and rounds
    the output once, so it sits inside the reference's own 16-bit error: bar 2e-3 for fp16 / 1e-2 for bf16 (output std ~0.3)."""
    g = gen(510 + groups)
    B, C, H, W, OC = 2, 256, 100, 136, 256
    x = torch.randn(B, C, H, W, generator=g).to(dtype)
    w = (torch.randn(OC, C // groups, 3, 3, generator=g) * (0.01 if groups == 1 else 0.2)).to(dtype)
    off = torch.randn(B, 18, H, W, generator=g).to(dtype)
    m = torch.rand(B, 9, H, W, generator=g).to(dtype)
    b = torch.randn(OC, generator=g).to(dtype)
    for use_mask in (False, True):
        y = vision_amd.deform_conv2d(x.to(DEV), off.to(DEV), w.to(DEV), b.to(DEV), (1, 1), (1, 1), (1, 1), m.to(DEV) if use_mask else None)
        assert y.dtype == dtype
        if O.load_reference():
            ref = torch.ops.torchvision.deform_conv2d(x.float(), w.float(), off.float(), m.float() if use_mask else torch.zeros(B, 1),
                                                      b.float(), 1, 1, 1, 1, 1, 1, groups, 1, use_mask).numpy()
        else:
            ref = O.deform_conv2d(x.float().numpy(), w.float().numpy(), off.float().numpy(), m.float().numpy(), b.float().numpy(),
                                  (1, 1), (1, 1), (1, 1), groups, 1, use_mask)
        np.testing.assert_allclose(y.float().cpu().numpy(), ref, rtol=tol, atol=tol, err_msg=f"mask={use_mask}")


@pytest.mark.parametrize("mode,aa", [("nearest", False), ("bilinear", False), ("bicubic", False), ("bilinear", True), ("bicubic", True)])
def test_resize_at_the_measured_shape(mode, aa):
    """8x3x1080x1920 -> 800x1422, the shape the resize kernels are timed on (tools/gpu_matrix.py): all five
    modes, the LDS-tiled kernels included, against installed-torch CPU F.interpolate (the arithmetic the reference's
    wrappers call, _geometry.py:344) at the 1e-4 bar."""
    import torch.nn.functional as F

    x = torch.rand(8, 3, 1080, 1920, generator=gen(21))
    kw = {} if mode == "nearest" else dict(align_corners=False, antialias=aa)
    want = F.interpolate(x, size=(800, 1422), mode=mode, **kw)
    with torch.no_grad():
        got = vision_amd.interpolate(x.to(DEV), size=(800, 1422), mode=mode, **kw).cpu()
    assert got.shape == want.shape
    assert float((got - want).abs().max()) <= TOL, (mode, aa)


def test_resize_config5_image_through_the_aten_override():
    """The resize of config 5: a 3x800x1333 image through torch.nn.functional.interpolate with the aten override on — the
    identity-size call GeneralizedRCNNTransform makes (models/detection/transform.py:65-72: scale_factor with
    recompute_scale_factor=True), an up- and a down-scale of the same image, bilinear like the transform and the FPN's nearest
    (ops/feature_pyramid_network.py:194) — against torch CPU; the call counter proves our kernel ran."""
    import torch.nn.functional as F

    img = torch.rand(1, 3, 800, 1333, generator=gen(22))
    d = img.to(DEV)
    was = vision_amd.override_aten_upsample(True)
    try:
        c0 = int(torch.ops.tvmi.aten_upsample_calls())
        cases = [dict(scale_factor=1.0, mode="bilinear", recompute_scale_factor=True, align_corners=False),
                 dict(size=(800, 1333), mode="bilinear", align_corners=False),
                 dict(scale_factor=0.6, mode="bilinear", recompute_scale_factor=True, align_corners=False),
                 dict(size=(1000, 1666), mode="bilinear", align_corners=False),
                 dict(size=(400, 667), mode="bicubic", align_corners=False, antialias=True),
                 dict(size=(1600, 2666), mode="nearest")]
        for kw in cases:
            want = F.interpolate(img, **kw)
            got = F.interpolate(d, **kw).cpu()
            assert got.shape == want.shape and float((got - want).abs().max()) <= TOL, kw
        assert int(torch.ops.tvmi.aten_upsample_calls()) - c0 == len(cases)
        # float64 is NOT taken over (resize.hip sums in float; a process-wide override must keep fp64 exact)
        c1 = int(torch.ops.tvmi.aten_upsample_calls())
        x64 = img[..., :96, :128].double()
        got = F.interpolate(x64.to(DEV), size=(150, 201), mode="bicubic", align_corners=False).cpu()
        assert int(torch.ops.tvmi.aten_upsample_calls()) == c1
        assert float((got - F.interpolate(x64, size=(150, 201), mode="bicubic", align_corners=False)).abs().max()) < 1e-12
    finally:
        vision_amd.override_aten_upsample(was)
