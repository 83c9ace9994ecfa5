"""oracle/oracle.py — TEST INFRASTRUCTURE ONLY: numpy binding of the C restatement."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(HERE, "liboracle.so")
_REF = os.path.join(HERE, "_ref", "libtv_ref_cpu.so")
_lib = None


def build(force=False):
    srcs = [os.path.join(HERE, f) for f in ("tvmi_oracle.c", "oracle_impl.inc")]
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["make", "-C", HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
    return _lib


def reference_available():
    return os.path.exists(_REF)


def load_reference():
    """Registers the REAL reference CPU kernels (oracle/_ref) on the CPU dispatch key of the
    `torchvision::` schemas owned by vision_amd's tvmi_torch.so.  Returns False if the
    prebuilt library is absent (e.g. never built in a checkout without /root/reference)."""
    import torch

    if getattr(load_reference, "_done", False):
        return True
    if not os.path.exists(_REF):
        return False
    import vision_amd  # noqa: F401  (owns the schema definitions)

    torch.ops.load_library(_REF)
    load_reference._done = True
    return True


def _sfx(a):
    if a.dtype == np.float32:
        return "f32", ctypes.c_float
    if a.dtype == np.float64:
        return "f64", ctypes.c_double
    raise TypeError(f"oracle supports float32/float64, got {a.dtype}")


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _c(a, dtype=None):
    return np.ascontiguousarray(a, dtype=dtype)


def stable_descending_order(scores):
    """aten::sort(stable=True, descending=True) order: ties keep the lower index first,
    NaN sorts as the largest value."""
    s = np.asarray(scores)
    key = np.where(np.isnan(s), np.inf, s)
    return np.argsort(-key, kind="stable").astype(np.int64)


def nms(dets, scores, iou_threshold, idxs=None):
    dets = _c(dets)
    sfx, _ = _sfx(dets)
    n = dets.shape[0]
    order = stable_descending_order(_c(scores))
    keep = np.empty(max(n, 1), dtype=np.int64)
    seg = None if idxs is None else _c(idxs, np.int64)
    fn = getattr(lib(), f"oracle_nms_{sfx}")
    fn.restype = ctypes.c_int64
    k = fn(_p(dets), _p(order), _p(seg) if seg is not None else None, ctypes.c_int64(n),
           ctypes.c_double(iou_threshold), _p(keep))
    return keep[:k].copy()


def batched_nms(boxes, scores, idxs, iou_threshold, device_type="cuda"):
    """torchvision/ops/boxes.py:57-126 for torch tensors: the reference's switch between the coordinate trick
    (:93-109: one nms() on `boxes + idxs * (boxes.max() + 1)`, taken up to 4000 elements on CPU / 100,000 on other
    devices, :83) and the per-category loop (:113-126).  The shift is done with torch arithmetic in the boxes' own dtype,
    exactly as the reference does it, so the shifted coordinates round the same way."""
    import torch

    if boxes.numel() == 0:
        return np.empty((0,), dtype=np.int64)
    if boxes.numel() > (4000 if device_type == "cpu" else 100_000):
        return nms(boxes.numpy(), scores.numpy(), iou_threshold, idxs.numpy())
    max_coordinate = boxes.max()
    offsets = idxs.to(boxes) * (max_coordinate + torch.tensor(1).to(boxes))
    return nms((boxes + offsets[:, None]).numpy(), scores.numpy(), iou_threshold)


def roi_align(x, rois, spatial_scale, ph, pw, sampling_ratio, aligned):
    x, rois = _c(x), _c(rois)
    sfx, _ = _sfx(x)
    N, C, H, W = x.shape
    K = rois.shape[0]
    out = np.zeros((K, C, ph, pw), dtype=x.dtype)
    getattr(lib(), f"oracle_roi_align_{sfx}")(_p(x), _p(rois), _p(out), K, C, H, W, ph, pw,
                                              ctypes.c_double(spatial_scale), int(sampling_ratio), int(aligned), 0)
    return out


def roi_align_backward(grad, rois, spatial_scale, ph, pw, N, C, H, W, sampling_ratio, aligned):
    grad, rois = _c(grad), _c(rois)
    sfx, _ = _sfx(grad)
    gin = np.zeros((N, C, H, W), dtype=grad.dtype)
    getattr(lib(), f"oracle_roi_align_{sfx}")(_p(grad), _p(rois), _p(gin), rois.shape[0], C, H, W, ph, pw,
                                              ctypes.c_double(spatial_scale), int(sampling_ratio), int(aligned), 1)
    return gin


def roi_pool(x, rois, spatial_scale, ph, pw):
    x, rois = _c(x), _c(rois)
    sfx, _ = _sfx(x)
    N, C, H, W = x.shape
    K = rois.shape[0]
    out = np.zeros((K, C, ph, pw), dtype=x.dtype)
    argmax = np.zeros((K, C, ph, pw), dtype=np.int32)
    getattr(lib(), f"oracle_roi_pool_{sfx}")(_p(x), _p(rois), _p(out), _p(argmax), K, C, H, W, ph, pw,
                                             ctypes.c_double(spatial_scale))
    return out, argmax


def roi_pool_backward(grad, rois, argmax, N, C, H, W):
    grad, rois, argmax = _c(grad), _c(rois), _c(argmax, np.int32)
    sfx, _ = _sfx(grad)
    K, _, ph, pw = grad.shape
    gin = np.zeros((N, C, H, W), dtype=grad.dtype)
    getattr(lib(), f"oracle_roi_pool_bwd_{sfx}")(_p(grad), _p(rois), _p(argmax), _p(gin), K, C, H, W, ph, pw)
    return gin


def ps_roi_align(x, rois, spatial_scale, ph, pw, sampling_ratio):
    x, rois = _c(x), _c(rois)
    sfx, _ = _sfx(x)
    N, C, H, W = x.shape
    K = rois.shape[0]
    co = C // (ph * pw)
    out = np.zeros((K, co, ph, pw), dtype=x.dtype)
    mapping = np.zeros((K, co, ph, pw), dtype=np.int32)
    getattr(lib(), f"oracle_ps_roi_align_{sfx}")(_p(x), _p(rois), _p(out), _p(mapping), K, C, H, W, ph, pw,
                                                 ctypes.c_double(spatial_scale), int(sampling_ratio), 0)
    return out, mapping


def ps_roi_align_backward(grad, rois, mapping, spatial_scale, ph, pw, sampling_ratio, N, C, H, W):
    grad, rois, mapping = _c(grad), _c(rois), _c(mapping, np.int32)
    sfx, _ = _sfx(grad)
    gin = np.zeros((N, C, H, W), dtype=grad.dtype)
    getattr(lib(), f"oracle_ps_roi_align_{sfx}")(_p(grad), _p(rois), _p(gin), _p(mapping), rois.shape[0], C, H, W, ph,
                                                 pw, ctypes.c_double(spatial_scale), int(sampling_ratio), 1)
    return gin


def ps_roi_pool(x, rois, spatial_scale, ph, pw):
    x, rois = _c(x), _c(rois)
    sfx, _ = _sfx(x)
    N, C, H, W = x.shape
    K = rois.shape[0]
    co = C // (ph * pw)
    out = np.zeros((K, co, ph, pw), dtype=x.dtype)
    mapping = np.zeros((K, co, ph, pw), dtype=np.int32)
    getattr(lib(), f"oracle_ps_roi_pool_{sfx}")(_p(x), _p(rois), _p(out), _p(mapping), K, C, H, W, ph, pw,
                                                ctypes.c_double(spatial_scale), 0)
    return out, mapping


def ps_roi_pool_backward(grad, rois, mapping, spatial_scale, ph, pw, N, C, H, W):
    grad, rois, mapping = _c(grad), _c(rois), _c(mapping, np.int32)
    sfx, _ = _sfx(grad)
    gin = np.zeros((N, C, H, W), dtype=grad.dtype)
    getattr(lib(), f"oracle_ps_roi_pool_{sfx}")(_p(grad), _p(rois), _p(gin), _p(mapping), rois.shape[0], C, H, W, ph,
                                                pw, ctypes.c_double(spatial_scale), 1)
    return gin


def deform_conv2d(x, weight, offset, mask, bias, stride, pad, dil, groups, offset_groups, use_mask):
    x, weight, offset, bias = _c(x), _c(weight), _c(offset), _c(bias)
    sfx, _ = _sfx(x)
    B, C, H, W = x.shape
    OC, _, kh, kw = weight.shape
    oh = (H + 2 * pad[0] - (dil[0] * (kh - 1) + 1)) // stride[0] + 1
    ow = (W + 2 * pad[1] - (dil[1] * (kw - 1) + 1)) // stride[1] + 1
    mask = _c(mask) if use_mask else np.zeros(1, dtype=x.dtype)
    out = np.zeros((B, OC, oh, ow), dtype=x.dtype)
    getattr(lib(), f"oracle_deform_conv2d_{sfx}")(_p(x), _p(weight), _p(offset), _p(mask), _p(bias), _p(out), B, C, H,
                                                  W, OC, kh, kw, stride[0], stride[1], pad[0], pad[1], dil[0], dil[1],
                                                  groups, offset_groups, int(use_mask))
    return out


def box_iou_rotated(b1, b2):
    b1, b2 = _c(b1), _c(b2)
    sfx, _ = _sfx(b1)
    out = np.zeros((b1.shape[0], b2.shape[0]), dtype=np.float32)
    getattr(lib(), f"oracle_box_iou_rotated_{sfx}")(_p(b1), _p(b2), _p(out), b1.shape[0], b2.shape[0])
    return out


_MODES = {"nearest": 0, "nearest-exact": 1, "bilinear": 2, "bicubic": 3}


def interpolate(x, size, mode, align_corners=False, antialias=False, scale=(-1.0, -1.0)):
    x = _c(x, np.float32)
    N, C, IH, IW = x.shape
    OH, OW = size
    out = np.zeros((N, C, OH, OW), dtype=np.float32)
    if antialias:
        lib().oracle_interpolate2d_aa(_p(x), _p(out), N * C, IH, IW, OH, OW, _MODES[mode] - 2, int(bool(align_corners)),
                                      ctypes.c_double(scale[0]), ctypes.c_double(scale[1]))
    else:
        lib().oracle_interpolate2d(_p(x), _p(out), N * C, IH, IW, OH, OW, _MODES[mode], int(bool(align_corners)),
                                   ctypes.c_double(scale[0]), ctypes.c_double(scale[1]))
    return out


def paste_masks_in_image(masks, boxes, img_shape, padding=1):
    """models/detection/roi_heads.py:486-500 restated per detection: expand_masks (:404-413, zero
    pad), expand_boxes (:378-395, float32 arithmetic op by op), int64 truncation, bilinear
    resize of the padded mask to the integer box size (paste_mask_in_image :416-437, via the C
    oracle's aten::upsample_bilinear2d restatement) and the clipped slice-assign."""
    masks = _c(masks, np.float32)
    boxes = _c(boxes, np.float32)
    im_h, im_w = int(img_shape[0]), int(img_shape[1])
    N, _, M, _ = masks.shape
    scale = np.float32(float(M + 2 * padding) / M)
    out = np.zeros((N, 1, im_h, im_w), dtype=np.float32)
    half = np.float32(0.5)
    for n in range(N):
        b = boxes[n]
        w_half = (b[2] - b[0]) * half * scale
        h_half = (b[3] - b[1]) * half * scale
        x_c = (b[2] + b[0]) * half
        y_c = (b[3] + b[1]) * half
        e = [int(np.trunc(v)) for v in (x_c - w_half, y_c - h_half, x_c + w_half, y_c + h_half)]
        w = max(e[2] - e[0] + 1, 1)
        h = max(e[3] - e[1] + 1, 1)
        padded = np.pad(masks[n, 0], padding)[None, None]
        m = interpolate(padded, (h, w), "bilinear")[0, 0]
        x_0, x_1 = max(e[0], 0), min(e[2] + 1, im_w)
        y_0, y_1 = max(e[1], 0), min(e[3] + 1, im_h)
        if x_1 > x_0 and y_1 > y_0:
            out[n, 0, y_0:y_1, x_0:x_1] = m[y_0 - e[1]:y_1 - e[1], x_0 - e[0]:x_1 - e[0]]
    return out


# ---- detector post-processing (python-level reference code restated in numpy float32)
_BBOX_XFORM_CLIP = float(np.log(1000.0 / 16))  # models/detection/_utils.py:141


def decode_boxes(rel_codes, boxes, weights):
    """BoxCoder.decode_single, models/detection/_utils.py:183-224 (float32, op by op)."""
    f = np.float32
    rel_codes, boxes = _c(rel_codes, f), _c(boxes, f)
    widths, heights = boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]
    ctr_x, ctr_y = boxes[:, 0] + f(0.5) * widths, boxes[:, 1] + f(0.5) * heights
    wx, wy, ww, wh = (f(w) for w in weights)
    dx, dy = rel_codes[:, 0::4] / wx, rel_codes[:, 1::4] / wy
    dw = np.minimum(rel_codes[:, 2::4] / ww, f(_BBOX_XFORM_CLIP))
    dh = np.minimum(rel_codes[:, 3::4] / wh, f(_BBOX_XFORM_CLIP))
    pcx, pcy = dx * widths[:, None] + ctr_x[:, None], dy * heights[:, None] + ctr_y[:, None]
    pw, ph = np.exp(dw) * widths[:, None], np.exp(dh) * heights[:, None]
    hw, hh = f(0.5) * pw, f(0.5) * ph
    return np.stack([pcx - hw, pcy - hh, pcx + hw, pcy + hh], axis=2)  # [R, C, 4]


def _clip(boxes, shape):
    """clip_boxes_to_image, ops/boxes.py:171-199."""
    h, w = shape
    out = boxes.copy()
    out[..., 0::2] = np.clip(boxes[..., 0::2], 0, np.float32(w))
    out[..., 1::2] = np.clip(boxes[..., 1::2], 0, np.float32(h))
    return out


def postprocess_detections(class_logits, box_regression, proposals, image_shapes, weights=(10.0, 10.0, 5.0, 5.0),
                           score_thresh=0.05, nms_thresh=0.5, detections_per_img=100):
    """RoIHeads.postprocess_detections, models/detection/roi_heads.py:680-737."""
    f = np.float32
    logits = _c(class_logits, f)
    C = logits.shape[1]
    pred = decode_boxes(box_regression, np.concatenate(proposals), weights)
    e = np.exp(logits - logits.max(1, keepdims=True))
    scores = e / e.sum(1, keepdims=True)
    out, start = [], 0
    for p, shape in zip(proposals, image_shapes):
        n = len(p)
        b = _clip(pred[start:start + n], shape)[:, 1:].reshape(-1, 4)
        s = scores[start:start + n, 1:].reshape(-1)
        lab = np.tile(np.arange(1, C), n)
        start += n
        sel = np.nonzero(s > f(score_thresh))[0]
        b, s, lab = b[sel], s[sel], lab[sel]
        sel = np.nonzero(((b[:, 2] - b[:, 0]) >= f(1e-2)) & ((b[:, 3] - b[:, 1]) >= f(1e-2)))[0]
        b, s, lab = b[sel], s[sel], lab[sel]
        keep = nms(b, s, nms_thresh, idxs=lab)[:detections_per_img]
        out.append((b[keep], s[keep], lab[keep]))
    return out


def filter_proposals(proposals, objectness, image_shapes, num_anchors_per_level, pre_nms_top_n, post_nms_top_n,
                     nms_thresh=0.7, score_thresh=0.0, min_size=1e-3):
    """RegionProposalNetwork.filter_proposals, models/detection/rpn.py:231-286."""
    f = np.float32
    proposals, objectness = _c(proposals, f), _c(objectness, f)
    B = proposals.shape[0]
    objectness = objectness.reshape(B, -1)
    out = []
    for i in range(B):
        idx, lvl, off = [], [], 0
        for l, n in enumerate(num_anchors_per_level):
            k = min(pre_nms_top_n, n)
            idx.append(stable_descending_order(objectness[i, off:off + n])[:k] + off)
            lvl.append(np.full(k, l))
            off += n
        idx, lvl = np.concatenate(idx), np.concatenate(lvl)
        b = _clip(proposals[i, idx], image_shapes[i])
        s = (f(1) / (f(1) + np.exp(-objectness[i, idx]))).astype(f)
        sel = np.nonzero(((b[:, 2] - b[:, 0]) >= f(min_size)) & ((b[:, 3] - b[:, 1]) >= f(min_size)))[0]
        b, s, lvl = b[sel], s[sel], lvl[sel]
        sel = np.nonzero(s >= f(score_thresh))[0]
        b, s, lvl = b[sel], s[sel], lvl[sel]
        keep = nms(b, s, nms_thresh, idxs=lvl)[:post_nms_top_n]
        out.append((b[keep], s[keep]))
    return out


def transform_images(images, min_size, max_size, mean, std, size_divisible=32, fixed_size=None):
    """GeneralizedRCNNTransform.forward, eval mode (models/detection/transform.py:119-255): normalize, resize by the
    reference's scale rule through the C oracle's bilinear restatement, zero-padded batch."""
    import math

    f = np.float32
    sizes, resized = [], []
    for img in images:
        img = _c(img, f)
        C, h, w = img.shape
        norm = (img - np.asarray(mean, f)[:, None, None]) / np.asarray(std, f)[:, None, None]
        if fixed_size is not None:
            oh, ow = int(fixed_size[1]), int(fixed_size[0])
        else:
            scale = min(float(min_size) / float(min(h, w)), float(max_size) / float(max(h, w)))
            oh, ow = int(math.floor(float(h) * scale)), int(math.floor(float(w) * scale))
        resized.append(interpolate(norm[None], (oh, ow), "bilinear")[0])
        sizes.append((oh, ow))
    stride = float(size_divisible)
    hp = int(math.ceil(float(max(s[0] for s in sizes)) / stride) * stride)
    wp = int(math.ceil(float(max(s[1] for s in sizes)) / stride) * stride)
    out = np.zeros((len(images), resized[0].shape[0], hp, wp), f)
    for i, r in enumerate(resized):
        out[i, :, : r.shape[1], : r.shape[2]] = r
    return out, sizes


def qnms(dets, scores, iou_threshold):
    """torchvision::qnms (quantized/cpu/qnms_kernel.cpp:22-120): integer boxes / scores; the boxes are evaluated as float32
    with the arithmetic of cpu/nms_kernel.cpp (the scale cancels), in the stable descending order of the integer scores."""
    d = np.asarray(dets).astype(np.float32)
    order = np.argsort(-np.asarray(scores).astype(np.int64), kind="stable").astype(np.int64)
    n = d.shape[0]
    areas = (d[:, 2] - d[:, 0]) * (d[:, 3] - d[:, 1])
    sup = np.zeros(n, dtype=bool)
    keep = []
    f = np.float32
    for _i in range(n):
        i = order[_i]
        if sup[i]:
            continue
        keep.append(i)
        js = order[_i + 1:]
        js = js[~sup[js]]
        xx1 = np.maximum(d[i, 0], d[js, 0])
        yy1 = np.maximum(d[i, 1], d[js, 1])
        xx2 = np.minimum(d[i, 2], d[js, 2])
        yy2 = np.minimum(d[i, 3], d[js, 3])
        inter = np.maximum(f(0), xx2 - xx1) * np.maximum(f(0), yy2 - yy1)
        with np.errstate(divide="ignore", invalid="ignore"):
            ovr = inter / (areas[i] + areas[js] - inter)
        sup[js[ovr.astype(np.float64) > iou_threshold]] = True
    return np.asarray(keep, dtype=np.int64)


def qroi_align(x, rois, input_scale, input_zero_point, rois_scale, rois_zero_point, spatial_scale, ph, pw, sampling_ratio, aligned):
    """torchvision::qroi_align (quantized/cpu/qroi_align_kernel.cpp:22-178), op by op in float32: RoIs dequantised on load,
    bilinear sums on the raw integers, one dequantisation, average, round-half-even re-quantisation, saturation.  Pure-python
    loops: small cases only."""
    f = np.float32
    x = np.asarray(x)
    rois = np.asarray(rois)
    _, C, H, W = x.shape
    K = rois.shape[0]
    info = np.iinfo(x.dtype)
    out = np.zeros((K, C, ph, pw), dtype=x.dtype)
    isc, rsc, ssc = f(input_scale), f(rois_scale), f(spatial_scale)
    off = f(0.5) if aligned else f(0.0)
    for k in range(K):
        sw, sh, ew, eh = [(f(rois[k, j]) - f(rois_zero_point)) * rsc * ssc - off for j in (1, 2, 3, 4)]
        rw, rh = ew - sw, eh - sh
        if not aligned:
            rw, rh = max(rw, f(1)), max(rh, f(1))
        bh, bw = rh / f(ph), rw / f(pw)
        gh = sampling_ratio if sampling_ratio > 0 else int(np.ceil(rh / f(ph)))
        gw = sampling_ratio if sampling_ratio > 0 else int(np.ceil(rw / f(pw)))
        count = f(max(gh * gw, 1))

        def axis(dim, start, binsz, grid, p, i):
            c = start + f(p) * binsz + f(i + 0.5) * binsz / f(grid)
            if c < -1.0 or c > dim:
                return None
            c = max(c, f(0))
            lo = int(c)
            if lo >= dim - 1:
                hi = lo = dim - 1
                c = f(lo)
            else:
                hi = lo + 1
            l_ = c - f(lo)
            return lo, hi, l_, f(1) - l_
        for c in range(C):
            plane = x[0, c].astype(np.float32)
            for p in range(ph):
                for q in range(pw):
                    val, sum_w = f(0), f(0)
                    for iy in range(gh):
                        ay = axis(H, sh, bh, gh, p, iy)
                        for ix in range(gw):
                            ax = axis(W, sw, bw, gw, q, ix)
                            if ay is None or ax is None:
                                continue
                            ylo, yhi, ly, hy = ay
                            xlo, xhi, lx, hx = ax
                            w1, w2, w3, w4 = hy * hx, hy * lx, ly * hx, ly * lx
                            val = f(val + f(f(f(w1 * plane[ylo, xlo] + w2 * plane[ylo, xhi]) + w3 * plane[yhi, xlo]) + w4 * plane[yhi, xhi]))
                            sum_w = f(sum_w + f(f(f(w1 + w2) + w3) + w4))
                    val = f(isc * f(val - f(f(input_zero_point) * sum_w)))
                    val = f(val / count)
                    inv = f(f(1.0) / isc)
                    qv = int(f(f(input_zero_point) + f(np.rint(f(val * inv)))))
                    out[k, c, p, q] = min(max(qv, info.min), info.max)
    return out
