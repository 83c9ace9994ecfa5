"""The shared-staging RoIAlign forward (vision_amd/csrc/roi_align_plane.hip) against the reference CPU kernel
(cpu/roi_align_kernel.cpp:18-115, roi_align_common.h:32-124), with every route forced through
`torch.ops.tvmi.set_option`:

  staged   staging_gain huge   -> every eligible RoI is served by the map-staging kernel (whole planes; row bands where
                                  the plane does not fit; min_band_rows lowered so that small test maps are cut too)
  per-roi  shared_staging 0    -> the per-RoI LDS-DMA wave kernel + mop-up only (the shipped default: it measured faster)
  auto     shared_staging 1    -> the device-side decision picks per level

All three must agree with the oracle at 1e-4 (fp32) and with EACH OTHER up to fp32 summation order (4e-6).  Also pinned here: what a NaN / Inf pixel does next to a zero-weight tap."""
import math

import numpy as np
import pytest
import torch

import vision_amd
from oracle import oracle as O
from helpers import gen, random_boxes, rois_for

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 1e-4


class route:
    """context manager: force one forward route, restore the defaults afterwards"""
    DEFAULTS = {"roi_align.shared_staging": 0, "roi_align.min_band_rows": 32, "roi_align.staging_gain_x16": 32,
                "roi_align.stage_whole_planes": 1, "roi_align.band_channels": 2}

    def __init__(self, name, min_band_rows=None):
        self.opts = dict(self.DEFAULTS)
        if name == "staged":
            self.opts["roi_align.shared_staging"] = 1
            self.opts["roi_align.staging_gain_x16"] = 1 << 20
        elif name == "auto":      # staging on, the device-side rule decides per level
            self.opts["roi_align.shared_staging"] = 1
        else:
            assert name == "per-roi"   # the shipped default
        if min_band_rows is not None:
            self.opts["roi_align.min_band_rows"] = min_band_rows

    def __enter__(self):
        for k, v in self.opts.items():
            assert torch.ops.tvmi.set_option(k, v)

    def __exit__(self, *a):
        for k, v in self.DEFAULTS.items():
            torch.ops.tvmi.set_option(k, v)


def _ref(x, rois, scale, P, aligned=False):
    if O.load_reference():
        return torch.ops.torchvision.roi_align(x, rois, scale, P, P, 2, aligned).numpy()
    return O.roi_align(x.numpy(), rois.numpy(), scale, P, P, 2, aligned)


def test_set_option_rejects_unknown_names():
    with pytest.raises(RuntimeError):
        torch.ops.tvmi.set_option("roi_align.no_such_switch", 1)


@pytest.mark.parametrize("P", [7, 14])
@pytest.mark.parametrize("aligned", [False, True])
def test_multiscale_all_routes_agree_with_the_reference(P, aligned):
    """FPN shapes of config 2 at 48 channels (not a multiple of any channel-group size in use), proposals of every size
    class incl. boxes clipped by / hanging over the image borders: whole-plane levels (P3-P5), the banded P2, RoIs too tall
    for a band (stay with the per-RoI kernel), all in one launch."""
    g = gen(70 + P)
    N, C = 3, 48
    feats = {str(i): torch.randn(N, C, 800 // s, 1344 // s, generator=g) for i, s in enumerate((4, 8, 16, 32))}
    boxes = []
    for _ in range(N):
        n = 300
        xy = torch.rand(n, 2, generator=g) * torch.tensor([1344.0, 800.0]) - 20.0
        side = torch.exp(torch.rand(n, generator=g) * (math.log(700.0) - math.log(8.0)) + math.log(8.0))
        asp = torch.exp((torch.rand(n, generator=g) * 2 - 1) * math.log(4.0))
        wh = torch.stack([side * asp.sqrt(), side / asp.sqrt()], 1)
        b = torch.cat([xy, xy + wh], 1)
        b[:40] = b[:40].clamp(min=0)
        b[:40, 2].clamp_(max=1344.0)
        b[:40, 3].clamp_(max=800.0)
        boxes.append(b)
    dfeats = {k: v.to(DEV) for k, v in feats.items()}
    dboxes = [b.to(DEV) for b in boxes]
    rois = torch.cat([torch.cat([torch.full((b.shape[0], 1), float(i)), b], 1) for i, b in enumerate(boxes)]).to(DEV)
    flist = [dfeats[str(i)] for i in range(4)]
    scales = [1 / 4, 1 / 8, 1 / 16, 1 / 32]
    outs = {}
    for name in ("staged", "per-roi", "auto"):
        with route(name), torch.no_grad():
            outs[name] = torch.ops.tvmi.multiscale_roi_align(flist, rois, scales, P, P, 2, aligned, 2, 5, 224.0, 4.0, 1e-6).cpu()
    # the staged kernel sums a bin as (column sums over y) + (the other column), the per-RoI kernels sample by sample:
    # same products, another association — equal up to fp32 rounding of sums of O(1) terms
    assert float((outs["staged"] - outs["per-roi"]).abs().max()) <= 4e-6 and float((outs["auto"] - outs["per-roi"]).abs().max()) <= 4e-6
    from vision_amd.poolers import LevelMapper
    levels = LevelMapper(2, 5)(boxes)
    r5 = rois.cpu()
    for lvl in range(4):
        sel = torch.nonzero(levels == lvl)[:, 0]
        assert sel.numel() > 0
        ref = _ref(feats[str(lvl)], r5[sel], scales[lvl], P, aligned)
        np.testing.assert_allclose(outs["staged"][sel].numpy(), ref, rtol=0, atol=TOL, err_msg=f"level {lvl}")


@pytest.mark.parametrize("H,W", [(1, 9), (2, 3), (3, 5), (25, 42), (13, 43), (50, 84), (7, 4), (64, 64), (100, 168), (200, 336), (96, 21)])
def test_schema_op_all_routes_on_awkward_maps(tv, H, W):
    """torchvision::roi_align (single level) through the staged route on awkward map sizes — widths not divisible by a
    16-byte piece, odd plane sizes (tail copy), maps of one row / two columns, planes that do not fit the LDS (bands) —
    with RoIs hugging every border and hanging outside; bands forced down to 8 rows so that small maps are cut as well."""
    g = gen(130 + H + W)
    N, C = 2, 21
    x = torch.randn(N, C, H, W, generator=g)
    k = 96
    x1 = torch.rand(k, generator=g) * W * 1.2 - 0.1 * W
    y1 = torch.rand(k, generator=g) * H * 1.2 - 0.1 * H
    bw = torch.rand(k, generator=g) ** 2 * W * 1.1
    bh = torch.rand(k, generator=g) ** 2 * H * 1.1
    rois = torch.stack([torch.randint(0, N, (k,), generator=g).float(), x1, y1, x1 + bw, y1 + bh], 1)
    rois[0, 1:] = torch.tensor([0.0, 0.0, float(W), float(H)])
    rois[1, 1:] = torch.tensor([W - 1.0, H - 1.0, float(W), float(H)])
    rois[2, 1:] = torch.tensor([W - 0.5, 0.0, W + 3.0, float(H)])
    rois[3, 1:] = torch.tensor([-2.0, -2.0, 0.4, 0.4])
    rois[4, 1:] = torch.tensor([W + 5.0, H + 5.0, W + 9.0, H + 9.0])     # every sample outside: zeros
    for P in (7, 14):
        for aligned in (False, True):
            ref = _ref(x, rois, 1.0, P, aligned)
            got = {}
            for name, mbr in (("staged", 8), ("staged", 32), ("per-roi", None), ("auto", None)):
                with route(name, mbr):
                    got[(name, mbr)] = tv.roi_align(x.to(DEV), rois.to(DEV), 1.0, P, P, 2, aligned).cpu()
                np.testing.assert_allclose(got[(name, mbr)].numpy(), ref, rtol=0, atol=TOL, err_msg=f"{name} {mbr} P={P} aligned={aligned}")
            assert float((got[("staged", 8)] - got[("per-roi", None)]).abs().max()) <= 4e-6


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 4e-3), (torch.bfloat16, 5e-3)])
def test_staged_route_16bit(tv, dtype, tol):
    """fp16 / bf16 maps (8 elements per 16-byte piece) through the staged route, single level with 16-bit RoIs and
    multi-scale with fp32 RoIs; equal to the per-RoI route bit for bit, within the reference's 16-bit bar of fp32."""
    g = gen(77)
    N, C, H, W = 2, 40, 50, 84
    x = torch.rand(N, C, H, W, generator=g).to(dtype)
    rois = rois_for(N, 150, W * 16, H * 16, 16, 500, g).to(dtype)
    res = {}
    for name in ("staged", "per-roi"):
        with route(name):
            res[name] = tv.roi_align(x.to(DEV), rois.to(DEV), 1 / 16, 7, 7, 2, False).cpu()
    assert res["staged"].dtype == dtype and float((res["staged"].float() - res["per-roi"].float()).abs().max()) <= tol
    ref = _ref(x.float(), rois.float(), 1 / 16, 7)
    np.testing.assert_allclose(res["staged"].float().numpy(), ref, rtol=tol, atol=tol)
    feats = [torch.rand(N, 24, 800 // s, 1344 // s, generator=g).to(dtype).to(DEV) for s in (4, 8, 16, 32)]
    boxes = torch.cat([torch.cat([torch.full((200, 1), float(i)), random_boxes(200, 1344, 800, 8, 600, g)], 1) for i in range(N)]).to(DEV)
    res = {}
    for name in ("staged", "per-roi"):
        with route(name), torch.no_grad():
            res[name] = torch.ops.tvmi.multiscale_roi_align(feats, boxes, [1 / 4, 1 / 8, 1 / 16, 1 / 32], 7, 7, 2, False, 2, 5, 224.0, 4.0, 1e-6)
    assert float((res["staged"].float() - res["per-roi"].float()).abs().max()) <= tol


def test_nonfinite_pixels_next_to_zero_weight_taps(tv):
    """VERDICT r02 weak 1d.  The reference reads, for a sample at c >= dim-1, the pixel dim-1 twice (x_high = x_low,
    roi_align_common.h:78-90) and never the pixel dim-2; a sample outside [-1, dim] is skipped (:60-73).  The fast kernels
    re-express the x edge on the pair (W-2, W-1) with factors (0, 1).  Pinned behaviour of EVERY forward kernel (map-staging,
    per-RoI LDS-DMA + mop-up, channels_last, generic): a NaN in a pixel the reference does not read never reaches the output
    (v_mul_legacy_f32 for the zero factor, the reference's own y_high = y_low row, skipped samples contribute an exact
    zero), and a NaN the reference does read gives NaN in exactly the same outputs."""
    g = gen(5)
    N, C, H, W = 1, 8, 20, 24
    x = torch.randn(N, C, H, W, generator=g)
    # RoI 0: right / bottom edge samples (c >= dim-1); RoI 1: partly outside the map (skipped samples); RoI 2: interior
    rois = torch.tensor([[0, 17.0, 12.0, 24.0, 20.0], [0, 18.0, -6.0, 30.0, 4.0], [0, 3.0, 3.0, 12.0, 11.0]])
    clean = _ref(x, rois, 1.0, 7)
    xb = x.clone()
    xb[:, :, :, W - 2] = float("nan")          # column W-2: read by RoI 0 / 1 only where their samples really lie in [W-3, W-1)
    xb[:, :, H - 2, :] = float("nan")          # row H-2: same for RoI 0's bottom bins
    ref = _ref(xb, rois, 1.0, 7)
    fin = ~np.isnan(ref)
    assert fin.any() and np.isnan(ref).any()
    # bins whose samples all sit on the last column / row are finite in the reference (it reads pixel dim-1 only)
    assert np.isfinite(ref[0, :, :, 6]).any() or np.isfinite(ref[0, :, 6, :]).any()
    results = {}
    for name in ("staged", "per-roi"):      # the map-staging kernel; the per-RoI LDS-DMA kernel + its mop-up (RoI 1 has skipped samples)
        with route(name):
            results[name] = tv.roi_align(xb.to(DEV), rois.to(DEV), 1.0, 7, 7, 2, False).cpu().numpy()
    # the channels_last kernel (reached through the multi-scale op; one level)
    xcl = xb.to(DEV).contiguous(memory_format=torch.channels_last)
    results["nhwc"] = torch.ops.tvmi.multiscale_roi_align([xcl], rois.to(DEV), [1.0], 7, 7, 2, False, 2, 5, 224.0, 4.0, 1e-6).cpu().numpy()
    # the kernel for other pooled shapes (5x5 here) skips like the reference by construction
    ref5 = torch.ops.torchvision.roi_align(xb, rois, 1.0, 5, 5, 2, False).numpy() if O.load_reference() else O.roi_align(xb.numpy(), rois.numpy(), 1.0, 5, 5, 2, False)
    got5 = tv.roi_align(xb.to(DEV), rois.to(DEV), 1.0, 5, 5, 2, False).cpu().numpy()
    assert np.array_equal(np.isnan(got5), np.isnan(ref5))
    for name, got in results.items():
        assert np.array_equal(np.isnan(got), np.isnan(ref)), f"{name}: NaN pattern differs from the reference CPU kernel"
        np.testing.assert_allclose(got[fin], ref[fin], rtol=0, atol=TOL, err_msg=name)
        # an interior RoI is untouched by the poisoned row / column
        np.testing.assert_allclose(got[2], clean[2], rtol=0, atol=TOL, err_msg=name)
