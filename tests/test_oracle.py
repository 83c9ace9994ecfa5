"""Pins the oracle: the plain-C restatement (oracle/tvmi_oracle.c) must reproduce
(a) the committed golden vectors generated from the reference's own CPU kernels,
(b) the reference itself (oracle/_ref) on fresh seeded inputs when it is available, and
(c) the known-answer vectors the reference's tests hold for this path."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from helpers import adversarial_nms_inputs, gen, golden, python_greedy_nms, rois_for


def test_nms_golden_bit_exact():
    g = golden("nms")
    for i in range(int(g["count"])):
        keep = O.nms(g[f"boxes{i}"], g[f"scores{i}"], float(g[f"thr{i}"]))
        assert np.array_equal(keep, g[f"keep{i}"]), f"case {i}"


def test_roi_ops_golden():
    g = golden("roi_ops")
    x, rois = g["x"], g["rois"]
    C = x.shape[1]
    for scale in (1.0, 0.5):
        for sr in (-1, 2):
            for aligned in (False, True):
                key = f"s{scale}_sr{sr}_a{int(aligned)}"
                y = O.roi_align(x, rois, scale, 5, 5, sr, aligned)
                np.testing.assert_array_equal(y, g["roi_align_" + key])
                gr = np.linspace(-1, 1, y.size, dtype=np.float32).reshape(y.shape)
                gin = O.roi_align_backward(gr, rois, scale, 5, 5, 2, C, 10, 10, sr, aligned)
                np.testing.assert_allclose(gin, g["roi_align_bwd_" + key], rtol=0, atol=1e-6)
            y, m = O.ps_roi_align(x, rois[:-1], scale, 5, 5, sr)
            np.testing.assert_array_equal(y, g[f"ps_roi_align_s{scale}_sr{sr}"])
            np.testing.assert_array_equal(m, g[f"ps_roi_align_map_s{scale}_sr{sr}"])
        y, a = O.roi_pool(x, rois, scale, 5, 5)
        np.testing.assert_array_equal(y, g[f"roi_pool_s{scale}"])
        np.testing.assert_array_equal(a, g[f"roi_pool_argmax_s{scale}"])
        y, m = O.ps_roi_pool(x, rois, scale, 5, 5)
        np.testing.assert_array_equal(y, g[f"ps_roi_pool_s{scale}"])
        np.testing.assert_array_equal(m, g[f"ps_roi_pool_map_s{scale}"])


def test_deform_conv2d_golden():
    g = golden("deform_conv2d")
    args = dict(stride=(2, 1), pad=(1, 0), dil=(2, 1), groups=2, offset_groups=3)
    out = O.deform_conv2d(g["x"], g["weight"], g["offset"], g["mask"], g["bias"], use_mask=True, **args)
    np.testing.assert_allclose(out, g["out_mask"], rtol=0, atol=1e-5)
    out = O.deform_conv2d(g["x"], g["weight"], g["offset"], None, g["bias"], use_mask=False, **args)
    np.testing.assert_allclose(out, g["out_nomask"], rtol=0, atol=1e-5)


def test_box_iou_rotated_golden():
    g = golden("box_iou_rotated")
    np.testing.assert_array_equal(O.box_iou_rotated(g["b1"], g["b2"]), g["iou"])
    np.testing.assert_array_equal(O.box_iou_rotated(g["b1"].astype(np.float64), g["b2"].astype(np.float64)), g["iou64"])


def test_box_iou_rotated_reference_known_answers():
    # analytic matrix of the reference's TestRotatedBoxIou.test_iou (test/test_ops.py:1848-1875)
    boxes = np.array([[0, 0, 10, 10, 45], [0, 0, 10, 10, 135], [0, 0, 10, 10, -45], [0, 0, 10, 10, -135],
                      [100, 100, 10, 10, 30], [50, 50, 20, 10, 45], [50, 50, 20, 10, 135], [50, 50, 20, 10, -135]],
                     dtype=np.float32)
    t = 1 / 3
    expected = np.array([[1, 1, 1, 1, 0, 0, 0, 0]] * 4 + [[0, 0, 0, 0, 1, 0, 0, 0], [0, 0, 0, 0, 0, 1, t, 1],
                                                            [0, 0, 0, 0, 0, t, 1, t], [0, 0, 0, 0, 0, 1, t, 1]], dtype=np.float32)
    for dt in (np.float32, np.float64):
        np.testing.assert_allclose(O.box_iou_rotated(boxes.astype(dt), boxes.astype(dt)), expected, atol=1e-4, rtol=1e-4)


def test_resize_golden_pins_torch_cpu():
    g = golden("resize")
    img = g["img"]
    for key in g.files:
        if key in ("img", "torch_version"):
            continue
        mode, size, flag = key.rsplit("_", 2)
        oh, ow = (int(v) for v in size.split("x"))
        out = O.interpolate(img, (oh, ow), mode, align_corners=flag == "ac1", antialias=flag == "aa1")
        np.testing.assert_allclose(out, g[key], rtol=0, atol=1e-5, err_msg=key)


def test_nms_matches_independent_greedy_loop():
    # same construction as the reference's test_nms_ref (test/test_ops.py:916-925)
    for thr in (0.2, 0.5, 0.8):
        for seed in range(3):
            boxes, scores = adversarial_nms_inputs(300, thr, gen(seed))
            keep = O.nms(boxes.numpy(), scores.numpy(), thr)
            assert np.array_equal(keep, python_greedy_nms(boxes, scores, thr).numpy())


def test_nms_segmented_equals_per_class_loop():
    g = gen(4)
    boxes = torch.rand(500, 4, generator=g) * 50
    boxes[:, 2:] += boxes[:, :2]
    scores = torch.rand(500, generator=g)
    idxs = torch.randint(0, 5, (500,), generator=g)
    keep = O.nms(boxes.numpy(), scores.numpy(), 0.4, idxs.numpy())
    mask = np.zeros(500, dtype=bool)
    for c in range(5):
        sel = np.nonzero(idxs.numpy() == c)[0]
        mask[sel[O.nms(boxes.numpy()[sel], scores.numpy()[sel], 0.4)]] = True
    expect = np.nonzero(mask)[0]
    expect = expect[np.argsort(-scores.numpy()[expect], kind="stable")]
    assert np.array_equal(keep, expect)


# ---------------------------------------------------------------- against the reference itself
def test_oracle_vs_reference_fresh_inputs(need_ref, tv):
    g = gen(21)
    for dt in (torch.float32, torch.float64):
        x = torch.randn(2, 18, 13, 11, generator=g).to(dt)
        rois = rois_for(2, 25, 22, 26, 2, 20, g, dt)
        rois[0, 1:] = torch.tensor([-5.0, -4.0, 40.0, 35.0], dtype=dt)
        for scale, sr, al in ((0.5, 2, False), (0.5, 0, True), (1.0, 3, True)):
            ref = tv.roi_align(x, rois, scale, 3, 3, sr, al).numpy()
            np.testing.assert_array_equal(O.roi_align(x.numpy(), rois.numpy(), scale, 3, 3, sr, al), ref)
            gr = torch.randn(ref.shape, generator=g).to(dt)
            refb = tv._roi_align_backward(gr, rois, scale, 3, 3, 2, 18, 13, 11, sr, al).numpy()
            np.testing.assert_allclose(O.roi_align_backward(gr.numpy(), rois.numpy(), scale, 3, 3, 2, 18, 13, 11, sr, al),
                                       refb, rtol=0, atol=1e-6)
        y, a = tv.roi_pool(x, rois, 0.5, 3, 3)
        yo, ao = O.roi_pool(x.numpy(), rois.numpy(), 0.5, 3, 3)
        np.testing.assert_array_equal(yo, y.numpy())
        np.testing.assert_array_equal(ao, a.numpy())
        y, m = tv.ps_roi_align(x, rois, 0.5, 3, 3, 2)
        yo, mo = O.ps_roi_align(x.numpy(), rois.numpy(), 0.5, 3, 3, 2)
        np.testing.assert_array_equal(yo, y.numpy())
        np.testing.assert_array_equal(mo, m.numpy())
        gr = torch.randn(y.shape, generator=g).to(dt)
        np.testing.assert_allclose(O.ps_roi_align_backward(gr.numpy(), rois.numpy(), mo, 0.5, 3, 3, 2, 2, 18, 13, 11),
                                   tv._ps_roi_align_backward(gr, rois, m, 0.5, 3, 3, 2, 2, 18, 13, 11).numpy(), atol=1e-6)
        y, m = tv.ps_roi_pool(x, rois, 0.5, 3, 3)
        yo, mo = O.ps_roi_pool(x.numpy(), rois.numpy(), 0.5, 3, 3)
        np.testing.assert_array_equal(yo, y.numpy())
        np.testing.assert_array_equal(mo, m.numpy())
        np.testing.assert_allclose(O.ps_roi_pool_backward(gr.numpy(), rois.numpy(), mo, 0.5, 3, 3, 2, 18, 13, 11),
                                   tv._ps_roi_pool_backward(gr, rois, m, 0.5, 3, 3, 2, 18, 13, 11).numpy(), atol=1e-6)
        boxes, scores = adversarial_nms_inputs(700, 0.5, g, dup=True)
        boxes, scores = boxes.to(dt), scores.to(dt)
        assert np.array_equal(O.nms(boxes.numpy(), scores.numpy(), 0.5), tv.nms(boxes, scores, 0.5).numpy())


def test_paste_masks_golden():
    """oracle restatement of paste_masks_in_image vs the reference python's output (tests/golden/detection.npz,
    oracle/gen_golden_detection.py).  Pasted rectangles must coincide exactly (integer box arithmetic); values
    within 1e-5 (bilinear association differs between aten's CPU kernel and the restatement)."""
    G = golden("detection")
    boxes, shape = G["paste_boxes"], tuple(int(v) for v in G["paste_shape"])
    for M, pad in ((28, 1), (14, 2), (7, 0)):
        want = G[f"paste_out_M{M}_p{pad}"]
        got = O.paste_masks_in_image(G[f"paste_masks_M{M}_p{pad}"], boxes, shape, padding=pad)
        assert got.shape == want.shape
        assert np.array_equal(got != 0, want != 0)
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-5)


def _det_golden():
    G = golden("detection")
    shapes = [tuple(int(v) for v in s) for s in G["det_shapes"]]
    props = [G[f"det_props{i}"] for i in range(len(shapes))]
    return G, shapes, props


def test_postprocess_detections_golden():
    """numpy restatement of RoIHeads.postprocess_detections vs the reference python's own output: same labels in the
    same order, boxes / scores within 1e-4 / 1e-6 (libm exp vs torch's vectorised exp)."""
    G, shapes, props = _det_golden()
    out = O.postprocess_detections(G["det_logits"], G["det_reg"], props, shapes, score_thresh=0.05, nms_thresh=0.5,
                                   detections_per_img=20)
    for i, (b, s, lab) in enumerate(out):
        assert np.array_equal(lab, G[f"det_labels{i}"])
        np.testing.assert_allclose(s, G[f"det_scores{i}"], rtol=0, atol=1e-6)
        np.testing.assert_allclose(b, G[f"det_boxes{i}"], rtol=0, atol=1e-4)


def test_filter_proposals_golden():
    G = golden("detection")
    shapes = [tuple(int(v) for v in s) for s in G["det_shapes"]]
    out = O.filter_proposals(G["rpn_proposals"], G["rpn_objectness"], shapes, [int(v) for v in G["rpn_levels"]], 60, 40,
                             nms_thresh=0.7, score_thresh=0.1)
    for i, (b, s) in enumerate(out):
        assert b.shape == G[f"rpn_boxes{i}"].shape
        np.testing.assert_allclose(s, G[f"rpn_scores{i}"], rtol=0, atol=1e-6)
        np.testing.assert_allclose(b, G[f"rpn_boxes{i}"], rtol=0, atol=1e-4)
    # decode restatement vs the reference's decoded proposals
    dec = O.decode_boxes(G["rpn_deltas"].reshape(-1, 4), G["rpn_anchors"].reshape(-1, 4), (1.0, 1.0, 1.0, 1.0))[:, 0]
    np.testing.assert_allclose(dec, G["rpn_proposals"].reshape(-1, 4), rtol=0, atol=1e-4)


def test_transform_images_golden():
    """numpy / C restatement of GeneralizedRCNNTransform.forward (eval) vs the reference's own output."""
    G = golden("detection")
    imgs = [G[f"xform_img{i}"] for i in range(4)]
    for tag, kw in (("a", dict(min_size=96, max_size=160)), ("b", dict(min_size=64, max_size=100)),
                    ("c", dict(min_size=50, max_size=80, fixed_size=(72, 56)))):
        out, sizes = O.transform_images(imgs, mean=[0.485, 0.456, 0.406], std=[0.229, 0.224, 0.225], **kw)
        assert [tuple(s) for s in G[f"xform_{tag}_sizes"]] == sizes
        assert out.shape == G[f"xform_{tag}_out"].shape
        np.testing.assert_allclose(out, G[f"xform_{tag}_out"], rtol=0, atol=2e-5)


def _quantized_case(dtype, g, K=30, C=5, H=19, W=23):
    info = torch.iinfo(dtype)
    lo, hi = max(info.min, -100), min(info.max, 200)
    x = torch.randint(lo, hi + 1, (1, C, H, W), generator=g).to(dtype)
    # RoIs in quantised image coordinates (rois_scale 0.5 -> coordinates 0 .. 2*extent), a few hanging outside / degenerate
    x1 = torch.randint(0, 2 * W, (K,), generator=g)
    y1 = torch.randint(0, 2 * H, (K,), generator=g)
    x2 = x1 + torch.randint(0, 2 * W, (K,), generator=g)
    y2 = y1 + torch.randint(0, 2 * H, (K,), generator=g)
    rois = torch.stack([torch.zeros(K, dtype=torch.int64), x1, y1, x2, y2], 1).clamp(max=min(info.max, 120)).to(dtype)
    return x, rois


def test_quantized_restatements_vs_reference(need_ref, tv):
    """oracle.qnms / oracle.qroi_align (numpy, op by op in float32) pinned to the reference's CPU kernels
    (quantized/cpu/qnms_kernel.cpp, qroi_align_kernel.cpp — part of oracle/_ref) bit for bit."""
    g = gen(33)
    for dtype in (torch.uint8, torch.int8, torch.int16, torch.int32):
        info = torch.iinfo(dtype)
        n = 300
        b = torch.randint(0, min(info.max, 100) - 40, (n, 2), generator=g)
        boxes = torch.cat([b, b + torch.randint(1, 40, (n, 2), generator=g)], 1).to(dtype)
        scores = torch.randint(max(info.min, -50), min(info.max, 120), (n,), generator=g).to(dtype)     # many ties: stable order matters
        for thr in (0.3, 0.5, 0.75):
            assert np.array_equal(O.qnms(boxes.numpy(), scores.numpy(), thr), tv.qnms(boxes, scores, thr).numpy()), (dtype, thr)
        x, rois = _quantized_case(dtype, g, K=8, C=2, H=11, W=13)
        zp = 3 if info.min < 0 else 120
        for (sr, aligned, scale) in ((2, False, 1.0), (0, True, 0.5), (3, False, 0.25)):
            ref = tv.qroi_align(x, rois, 0.07, zp, 0.5, 0, scale, 3, 4, sr, aligned).numpy()
            got = O.qroi_align(x.numpy(), rois.numpy(), 0.07, zp, 0.5, 0, scale, 3, 4, sr, aligned)
            np.testing.assert_array_equal(got, ref, err_msg=f"{dtype} sr={sr} aligned={aligned}")
