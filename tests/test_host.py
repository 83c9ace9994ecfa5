"""Host-side mirror of the reference interface (vision_amd/{boxes,roi_ops,poolers,deform_conv,
resize}.py): argument handling, strategy switches and error behaviour.  Compute on CPU tensors
goes through the REAL reference CPU kernels (oracle/_ref) — tests that need compute are skipped
when it is absent; nothing here needs a GPU."""
import math

import numpy as np
import pytest
import torch

import vision_amd
from vision_amd import boxes as B
from vision_amd.poolers import LevelMapper, _infer_scale, _setup_scales
from vision_amd.resize import _compute_resized_output_size
from helpers import gen, random_boxes


def test_box_iou_known_answers():
    # reference goldens: test/test_ops.py:1652-1658, 1716-1717
    int_boxes = torch.tensor([[0, 0, 100, 100], [0, 0, 50, 50], [200, 200, 300, 300], [0, 0, 25, 25]])
    int_boxes2 = torch.tensor([[0, 0, 100, 100], [0, 0, 50, 50], [200, 200, 300, 300]])
    expect = torch.tensor([[1.0, 0.25, 0.0], [0.25, 1.0, 0.0], [0.0, 0.0, 1.0], [0.0625, 0.25, 0.0]])
    for dt in (torch.int16, torch.int32, torch.int64):
        torch.testing.assert_close(B.box_iou(int_boxes.to(dt), int_boxes2.to(dt)), expect, atol=1e-4, rtol=0)
    fb = torch.tensor([[285.3538, 185.5758, 1193.5110, 851.4551], [285.1472, 188.7374, 1192.4984, 851.0669],
                       [279.2440, 197.9812, 1189.4746, 849.2019]])
    fexp = torch.tensor([[1.0, 0.9933, 0.9673], [0.9933, 1.0, 0.9737], [0.9673, 0.9737, 1.0]])
    torch.testing.assert_close(B.box_iou(fb, fb), fexp, atol=1e-3, rtol=0)
    torch.testing.assert_close(B.box_iou(fb.half(), fb.half()), fexp, atol=2e-3, rtol=0)


def test_box_helpers():
    b = torch.tensor([[-5.0, 3.0, 120.0, 40.0], [10.0, 10.0, 11.0, 30.0]])
    c = B.clip_boxes_to_image(b, (32, 100))
    assert c.tolist() == [[0.0, 3.0, 100.0, 32.0], [10.0, 10.0, 11.0, 30.0]]
    assert B.remove_small_boxes(b, 2.0).tolist() == [0]
    assert B.box_area(b).tolist() == [125.0 * 37.0, 20.0]


def test_nms_error_messages(need_ref):
    with pytest.raises(RuntimeError, match="boxes should be a 2d tensor"):
        vision_amd.nms(torch.rand(4), torch.rand(3), 0.5)
    with pytest.raises(RuntimeError, match="boxes should have 4 elements in dimension 1"):
        vision_amd.nms(torch.rand(3, 5), torch.rand(5), 0.5)
    with pytest.raises(RuntimeError, match="scores should be a 1d tensor"):
        vision_amd.nms(torch.rand(3, 4), torch.rand(3, 2), 0.5)
    with pytest.raises(RuntimeError, match="boxes and scores should have same number of elements"):
        vision_amd.nms(torch.rand(3, 4), torch.rand(4), 0.5)


def test_batched_nms_strategies_agree(need_ref):
    # reference test_batched_nms_implementations (test/test_ops.py:1024-1048)
    for seed in range(5):
        g = gen(seed)
        boxes = torch.cat((torch.rand(1000, 2, generator=g), torch.rand(1000, 2, generator=g) + 10), dim=1)
        scores = torch.rand(1000, generator=g)
        idxs = torch.randint(0, 4, (1000,), generator=g)
        a = B._batched_nms_vanilla(boxes, scores, idxs, 0.9)
        b = B._batched_nms_coordinate_trick(boxes, scores, idxs, 0.9)
        torch.testing.assert_close(a, b)
    empty = torch.empty((0, 4))
    assert vision_amd.batched_nms(empty, torch.empty(0), torch.empty(0, dtype=torch.int64), 0.5).numel() == 0


def test_batched_nms_switch(need_ref, monkeypatch):
    calls = []
    monkeypatch.setattr(B, "_batched_nms_vanilla", lambda *a: calls.append("vanilla") or torch.empty(0))
    monkeypatch.setattr(B, "_batched_nms_coordinate_trick", lambda *a: calls.append("trick") or torch.empty(0))
    vision_amd.batched_nms(torch.rand(1000, 4), torch.rand(1000), torch.zeros(1000), 0.5)
    vision_amd.batched_nms(torch.rand(1001, 4), torch.rand(1001), torch.zeros(1001), 0.5)
    assert calls == ["trick", "vanilla"]  # CPU threshold: 4000 elements


def test_roi_wrappers_accept_box_lists_and_check_shapes(need_ref):
    g = gen(1)
    x = torch.randn(2, 4, 16, 16, generator=g)
    b0, b1 = random_boxes(3, 32, 32, 2, 20, g), random_boxes(2, 32, 32, 2, 20, g)
    rois = torch.cat([torch.cat([torch.zeros(3, 1), b0], 1), torch.cat([torch.ones(2, 1), b1], 1)])
    for fn, kw in ((vision_amd.roi_align, dict(sampling_ratio=2, aligned=True)), (vision_amd.roi_pool, {})):
        torch.testing.assert_close(fn(x, [b0, b1], 3, 0.5, **kw), fn(x, rois, (3, 3), 0.5, **kw))
    with pytest.raises(AssertionError):
        vision_amd.roi_align(x, torch.rand(3, 4), 3)
    with pytest.raises(AssertionError):
        vision_amd.roi_align(x, [torch.rand(3, 5)], 3)
    m = vision_amd.RoIAlign((3, 3), 0.5, 2, aligned=True)
    torch.testing.assert_close(m(x, rois), vision_amd.roi_align(x, rois, 3, 0.5, 2, True))
    assert "RoIAlign(output_size=(3, 3)" in repr(m)


def test_roi_align_autograd_and_autocast_wrappers(need_ref):
    g = gen(2)
    x = torch.randn(1, 3, 8, 8, generator=g, dtype=torch.float64, requires_grad=True)
    rois = torch.tensor([[0, 1.0, 1.5, 6.0, 7.0], [0, 0.0, 0.0, 3.0, 3.0]], dtype=torch.float64)
    assert torch.autograd.gradcheck(lambda t: vision_amd.roi_align(t, rois, 2, 1.0, 2, True), (x,), nondet_tol=1e-5)
    assert torch.autograd.gradcheck(lambda t: vision_amd.ps_roi_align(torch.cat([t] * 4, 1), rois, 2, 1.0, 2), (x,),
                                    nondet_tol=1e-5)
    y = vision_amd.roi_align(x, rois, 2, 1.0, 2, True)
    (gy,) = torch.autograd.grad((y * y).sum(), x, create_graph=True)
    with pytest.raises(RuntimeError, match="double backwards on roi_align not supported"):
        gy.sum().backward()
    with torch.autocast("cpu", dtype=torch.bfloat16):
        out = vision_amd.roi_align(x.detach().bfloat16(), rois.bfloat16(), 2, 1.0, 2, False)
    assert out.dtype == torch.bfloat16


def test_deform_conv2d_wrapper(need_ref):
    g = gen(3)
    x = torch.rand(2, 6, 7, 6, generator=g)
    w = torch.randn(4, 3, 3, 3, generator=g)
    off = torch.randn(2, 2 * 9, 5, 4, generator=g)
    y = vision_amd.deform_conv2d(x, off, w)
    assert y.shape == (2, 4, 5, 4)
    # zero offsets == plain convolution
    y0 = vision_amd.deform_conv2d(x, torch.zeros_like(off), w, padding=0)
    torch.testing.assert_close(y0, torch.nn.functional.conv2d(x, w, groups=2), atol=1e-4, rtol=1e-4)
    with pytest.raises(RuntimeError, match="the shape of the offset tensor at dimension 1 is not valid"):
        vision_amd.deform_conv2d(x, off[:, :5], w)
    with pytest.raises(RuntimeError, match="mask.shape\\[1\\] is not valid"):
        vision_amd.deform_conv2d(x, off, w, mask=torch.rand(2, 5, 5, 4))
    m = vision_amd.DeformConv2d(6, 4, 3, groups=2)
    assert m(x, off).shape == (2, 4, 5, 4) and "groups=2" in repr(m)
    with pytest.raises(ValueError):
        vision_amd.DeformConv2d(5, 4, 3, groups=2)


def test_level_mapper_and_scales():
    lm = LevelMapper(2, 5)
    boxes = torch.tensor([[0, 0, 10, 10], [0, 0, 112, 112], [0, 0, 224, 224], [0, 0, 448, 448], [0, 0, 2000, 2000]],
                         dtype=torch.float32)
    assert lm([boxes]).tolist() == [0, 1, 2, 3, 3]
    feats = [torch.empty(1, 2, 200, 336), torch.empty(1, 2, 100, 168), torch.empty(1, 2, 50, 84), torch.empty(1, 2, 25, 42)]
    scales, mapper = _setup_scales(feats, [(800, 1344)], 224, 4)
    assert scales == [0.25, 0.125, 0.0625, 0.03125] and (mapper.k_min, mapper.k_max) == (2, 5)
    assert _infer_scale(torch.empty(1, 1, 13, 21), (800, 1333)) == 2 ** -6
    with pytest.raises(ValueError):
        _setup_scales(feats, [], 224, 4)


def test_multiscale_roi_align_matches_manual_levels(need_ref):
    g = gen(5)
    feats = {"0": torch.randn(2, 4, 64, 64, generator=g), "1": torch.randn(2, 4, 32, 32, generator=g),
             "skip": torch.randn(2, 4, 8, 8, generator=g), "2": torch.randn(2, 4, 16, 16, generator=g)}
    boxes = [random_boxes(9, 256, 256, 4, 250, g), random_boxes(7, 256, 256, 4, 250, g)]
    pool = vision_amd.MultiScaleRoIAlign(["0", "1", "2"], 3, 2)
    out = pool(feats, boxes, [(256, 256), (250, 256)])
    assert out.shape == (16, 4, 3, 3) and pool.scales == [0.25, 0.125, 0.0625]
    lv = pool.map_levels(boxes)
    rois = torch.cat([torch.cat([torch.full((len(b), 1), float(i)), b], 1) for i, b in enumerate(boxes)])
    for k in range(16):
        lvl = int(lv[k])
        exp = vision_amd.roi_align(feats[str(lvl)], rois[k : k + 1], 3, pool.scales[lvl], 2)
        torch.testing.assert_close(out[k : k + 1], exp)


def test_resize_output_size_rules():
    assert _compute_resized_output_size((480, 640), [800]) == [800, 1066]
    assert _compute_resized_output_size((480, 640), 800, max_size=1000) == [750, 1000]
    assert _compute_resized_output_size((640, 480), [800], max_size=1333) == [1066, 800]
    assert _compute_resized_output_size((480, 640), [100, 50]) == [100, 50]
    assert _compute_resized_output_size((480, 640), None, max_size=320) == [240, 320]
    with pytest.raises(ValueError):
        _compute_resized_output_size((480, 640), [800], max_size=700)


def test_interpolate_argument_validation():
    x = torch.rand(1, 1, 4, 4)
    with pytest.raises(ValueError):
        vision_amd.interpolate(x, size=(2, 2), scale_factor=2.0)
    with pytest.raises(ValueError):
        vision_amd.interpolate(x)
    with pytest.raises(ValueError):
        vision_amd.interpolate(x, size=(2, 2), mode="nearest", align_corners=True)
    with pytest.raises(ValueError):
        vision_amd.interpolate(x, size=(2, 2), mode="nearest", antialias=True)
    with pytest.raises(NotImplementedError):
        vision_amd.interpolate(torch.rand(1, 4, 4), size=(2, 2))
    with pytest.raises(NotImplementedError):
        vision_amd.interpolate(x, size=(2, 2), mode="area")


def test_cpu_tensors_without_reference_have_no_silent_fallback():
    # our library registers CUDA kernels only: a CPU call either reaches the reference oracle
    # (when tests loaded it) or raises — it never silently computes something else.
    from oracle import oracle as O

    if O.reference_available():
        pytest.skip("reference CPU kernels are loaded in this process")
    with pytest.raises((NotImplementedError, RuntimeError)):
        vision_amd.nms(torch.rand(3, 4), torch.rand(3), 0.5)


def test_transform_size_rule_matches_reference_golden():
    """`vision_amd.transform.resized_size` (host arithmetic of GeneralizedRCNNTransform) vs the image_sizes the
    reference module produced (tests/golden/detection.npz)."""
    from helpers import golden
    from vision_amd.transform import resized_size

    G = golden("detection")
    shapes = [G[f"xform_img{i}"].shape[-2:] for i in range(4)]
    for tag, kw in (("a", dict(min_size=96, max_size=160)), ("b", dict(min_size=64, max_size=100)),
                    ("c", dict(min_size=50, max_size=80, fixed_size=(72, 56)))):
        got = [resized_size(int(h), int(w), **kw) for h, w in shapes]
        assert got == [tuple(int(v) for v in s) for s in G[f"xform_{tag}_sizes"]]


def test_nms_band_test_never_contradicts_the_exact_predicate():
    """The NMS tile kernels (vision_amd/csrc/nms.hip: thr_band / suppression_row_pair) decide a pair without the
    division:  t = fma(-c, union, inter), ru = r * union;  t > ru -> suppressed, t < -ru -> not suppressed, and fall
    back to the reference's `(double)(inter / union) > thr` otherwise (also for unions / thresholds outside the
    fast-path range).  Emulate that float32 arithmetic here (the fma through float64: 24x24-bit product exact, one
    rounding of the sum, one to float — off from a true fma by at most a double-rounding tie) and check, on random and
    on adversarial (within a few ulps of the threshold) pairs, that a decided pair always agrees with the exact
    predicate — i.e. the margins of the band are sufficient."""
    f = np.float32
    rng = np.random.default_rng(0)

    def band(thr):
        if not (thr >= 1.0 / 1048576.0 and thr <= 1048576.0):
            return f(0), f(np.inf)
        d = f(thr)
        if float(d) > thr:
            d = np.nextafter(d, f(-np.inf))
        u = np.nextafter(d, f(np.inf))
        hi = np.nextafter(f(float(u) * (1.0 + 1.0 / 1048576.0)), f(np.inf))
        lo = np.nextafter(f(float(d) * (1.0 - 1.0 / 1048576.0)), f(-np.inf))
        return f(f(0.5) * f(hi + lo)), f(f(0.75) * f(hi - lo))

    thrs = [0.5, 0.3, 0.7, 1.0 / 3.0, 0.05, 0.95, 1.0, 1e-5, float(np.nextafter(f(0.5), f(1))), 0.5000000001, 0.4999999999]
    for thr in thrs:
        c, r = band(thr)
        assert np.isfinite(r) and r > 0
        for scale in (1e4, 2.0 ** -50, 2.0 ** 50):          # unions across the fast-path range [2^-60, 2^60]
            union = ((rng.random(400_000) + 1e-3) * scale).astype(f)
            # inter spread over [0, union] plus a cloud concentrated within a few ulps of thr * union
            inter_a = (rng.random(200_000).astype(f) * union[:200_000]).astype(f)
            base = (f(thr) * union[200_000:]).astype(f)
            steps = rng.integers(-40, 41, size=base.shape)
            inter_b = base.copy()
            for _ in range(40):
                up = steps > 0
                dn = steps < 0
                inter_b = np.where(up, np.nextafter(inter_b, f(np.inf)), np.where(dn, np.nextafter(inter_b, f(-np.inf)), inter_b))
                steps = steps - np.sign(steps)
            inter = np.concatenate([inter_a, np.maximum(inter_b, f(0))]).astype(f)
            with np.errstate(over="ignore", invalid="ignore"):
                t = (inter.astype(np.float64) - float(c) * union.astype(np.float64)).astype(f)
                ru = (r * union).astype(f)
                sure_t = t > ru
                sure_f = t < -ru
                exact = (inter / union).astype(f).astype(np.float64) > thr
            assert not np.any(sure_t & sure_f)
            assert np.all(exact[sure_t]), (thr, scale)
            assert not np.any(exact[sure_f]), (thr, scale)
            # and the band is narrow: almost everything is decided without the division
            assert (sure_t | sure_f).mean() > 0.45
    assert band(1e-7) == (f(0), f(np.inf)) and band(float("nan"))[1] == f(np.inf)    # outside the range: never decided


def test_distance_and_complete_box_iou_known_answers():
    """Host mirrors of distance_box_iou / complete_box_iou reproduce the reference's known-answer matrices
    (test/test_ops.py:1781-1788, 1811-1818), integer boxes included (tensor-math form on CPU)."""
    import torch
    import vision_amd

    ints1 = torch.tensor([[0, 0, 100, 100], [0, 0, 50, 50], [200, 200, 300, 300], [0, 0, 25, 25]])
    ints2 = torch.tensor([[0, 0, 100, 100], [0, 0, 50, 50], [200, 200, 300, 300]])
    int_expected = torch.tensor([[1.0, 0.1875, -0.4444], [0.1875, 1.0, -0.5625], [-0.4444, -0.5625, 1.0], [-0.0781, 0.1875, -0.6267]])
    for dt in (torch.int16, torch.int32, torch.int64):
        assert torch.allclose(vision_amd.distance_box_iou(ints1.to(dt), ints2.to(dt)), int_expected, atol=1e-4)
        assert torch.allclose(vision_amd.complete_box_iou(ints1.to(dt), ints2.to(dt)), int_expected, atol=1e-4)
