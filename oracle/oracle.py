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
