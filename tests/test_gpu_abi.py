"""Compute calls STRAIGHT through the C ABI (`-m gpu`): ctypes on libtvmi_kernels.so with raw device pointers, sizes
and a stream handle — no dispatcher glue in between — compared with the oracle.  This is the boundary a non-torch host
 would bind; torch is used here only to own the device memory."""
import ctypes

import numpy as np
import pytest
import torch

import vision_amd
from oracle import oracle as O
from helpers import adversarial_nms_inputs, gen, rois_for

pytestmark = pytest.mark.gpu
DEV = "cuda"
TVMI_F32 = 0
vp = ctypes.c_void_p
i64 = ctypes.c_int64


def _lib():
    lib = vision_amd._loader.kernels()
    lib.tvmi_last_error.restype = ctypes.c_char_p
    lib.tvmi_nms_workspace_bytes.restype = ctypes.c_size_t
    lib.tvmi_nms_workspace_bytes.argtypes = [i64]
    lib.tvmi_nms.restype = ctypes.c_int
    lib.tvmi_nms.argtypes = [vp, vp, vp, i64, ctypes.c_double, ctypes.c_int, vp, ctypes.c_size_t, vp, vp, vp]
    lib.tvmi_roi_align_forward.restype = ctypes.c_int
    lib.tvmi_roi_align_forward.argtypes = [vp, vp, vp, ctypes.c_int] + [i64] * 7 + [ctypes.c_double, i64, ctypes.c_int, vp,
                                                                                  ctypes.c_size_t, vp]
    lib.tvmi_roi_align_backward_workspace_bytes.restype = ctypes.c_size_t
    lib.tvmi_roi_align_backward_workspace_bytes.argtypes = [i64] * 4
    lib.tvmi_roi_align_backward_overwrites.restype = ctypes.c_int
    lib.tvmi_roi_align_backward_overwrites.argtypes = [ctypes.c_int] + [i64] * 10 + [ctypes.c_size_t]
    lib.tvmi_roi_align_backward.restype = ctypes.c_int
    lib.tvmi_roi_align_backward.argtypes = [vp, vp, vp, ctypes.c_int] + [i64] * 7 + [ctypes.c_double, i64, ctypes.c_int] + [i64] * 4 + [
        vp, ctypes.c_size_t, vp]
    return lib


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def test_c_abi_nms_on_raw_device_pointers():
    lib = _lib()
    for n, thr, seed in ((1000, 0.5, 0), (5000, 0.3, 1)):
        boxes, scores = adversarial_nms_inputs(n, thr, gen(seed))
        order = torch.from_numpy(O.stable_descending_order(scores.numpy()))
        d_boxes, d_order = boxes.to(DEV), order.to(DEV)
        wb = lib.tvmi_nms_workspace_bytes(n)
        ws = torch.empty(wb, dtype=torch.uint8, device=DEV)
        keep = torch.empty(n, dtype=torch.int64, device=DEV)
        num = torch.zeros(1, dtype=torch.int64, device=DEV)
        st = lib.tvmi_nms(d_boxes.data_ptr(), d_order.data_ptr(), None, n, thr, TVMI_F32, ws.data_ptr(), wb, keep.data_ptr(),
                          num.data_ptr(), _stream())
        assert st == 0, lib.tvmi_last_error()
        torch.cuda.synchronize()
        got = keep[: int(num)].cpu().numpy()
        assert np.array_equal(got, O.nms(boxes.numpy(), scores.numpy(), thr))
    # argument errors come back as a status + message, never as an exception or a crash
    assert lib.tvmi_nms(None, None, None, 10, 0.5, TVMI_F32, None, 0, None, None, _stream()) != 0
    assert lib.tvmi_last_error()


def test_c_abi_score_sort_and_blocking_nms_of_a_large_list():
    """The two entries a non-torch host would chain for a large list: tvmi_sort_scores_desc_large (the processing order) and
    tvmi_nms_blocking (re-plans on the survivors, one host synchronisation per re-plan) — raw pointers only, against the
    reference CPU algorithm; tvmi_nms (stream-ordered, never synchronises) must give the same list."""
    lib = _lib()
    lib.tvmi_sort_scores_desc_workspace_bytes.restype = ctypes.c_size_t
    lib.tvmi_sort_scores_desc_workspace_bytes.argtypes = [i64]
    lib.tvmi_sort_scores_desc_large.restype = ctypes.c_int
    lib.tvmi_sort_scores_desc_large.argtypes = [vp, i64, vp, vp, ctypes.c_size_t, vp]
    lib.tvmi_nms_blocking.restype = ctypes.c_int
    lib.tvmi_nms_blocking.argtypes = lib.tvmi_nms.argtypes
    g = gen(5)
    n, thr = 40_000, 0.5
    xy = torch.rand(n, 2, generator=g) * 600
    boxes = torch.cat([xy, xy + 4 + torch.rand(n, 2, generator=g) * 90], 1)
    scores = (torch.rand(n, generator=g) * 4096).floor() / 4096      # ties: the order must be the stable one
    d_boxes, d_scores = boxes.to(DEV), scores.to(DEV)
    order = torch.empty(n, dtype=torch.int64, device=DEV)
    sb = lib.tvmi_sort_scores_desc_workspace_bytes(n)
    sws = torch.empty(sb, dtype=torch.uint8, device=DEV)
    assert sws.data_ptr() % 256 == 0
    st = lib.tvmi_sort_scores_desc_large(d_scores.data_ptr(), n, order.data_ptr(), sws.data_ptr(), sb, _stream())
    assert st == 0, lib.tvmi_last_error()
    torch.cuda.synchronize()
    assert np.array_equal(order.cpu().numpy(), O.stable_descending_order(scores.numpy()))
    want = O.nms(boxes.numpy(), scores.numpy(), thr)
    wb = lib.tvmi_nms_workspace_bytes(n)
    ws = torch.empty(wb, dtype=torch.uint8, device=DEV)
    for entry in (lib.tvmi_nms_blocking, lib.tvmi_nms):
        keep = torch.empty(n, dtype=torch.int64, device=DEV)
        num = torch.zeros(1, dtype=torch.int64, device=DEV)
        st = entry(d_boxes.data_ptr(), order.data_ptr(), None, n, thr, TVMI_F32, ws.data_ptr(), wb, keep.data_ptr(), num.data_ptr(),
                   _stream())
        assert st == 0, lib.tvmi_last_error()
        torch.cuda.synchronize()
        assert np.array_equal(keep[: int(num)].cpu().numpy(), want)


def test_c_abi_roi_align_forward_and_owner_backward():
    lib = _lib()
    g = gen(3)
    N, C, H, W, K, P = 2, 24, 37, 53, 90, 7
    x = torch.randn(N, C, H, W, generator=g)
    rois = rois_for(N, K, W * 8, H * 8, 8, 250, g)
    d_x, d_rois = x.to(DEV), rois.to(DEV)
    out = torch.empty(K, C, P, P, device=DEV)
    flags = torch.empty(K, dtype=torch.int32, device=DEV)
    st = lib.tvmi_roi_align_forward(d_x.data_ptr(), d_rois.data_ptr(), out.data_ptr(), TVMI_F32, N, C, H, W, K, P, P, 1 / 8, 2, 0,
                                    flags.data_ptr(), K * 4, _stream())
    assert st == 0, lib.tvmi_last_error()
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.cpu().numpy(), O.roi_align(x.numpy(), rois.numpy(), 1 / 8, P, P, 2, False), rtol=0, atol=1e-4)
    # backward, tile-owner regime: the library says it overwrites, so the buffer is handed over full of garbage
    grad = torch.randn(K, C, P, P, generator=g)
    d_grad = grad.to(DEV)
    wb = lib.tvmi_roi_align_backward_workspace_bytes(N, K, P, P)
    assert wb > 0
    assert lib.tvmi_roi_align_backward_overwrites(TVMI_F32, N, C, H, W, K, P, P, P * P, P, 1, wb) == 1
    assert lib.tvmi_roi_align_backward_overwrites(TVMI_F32, N, C, H, W, K, 5, 5, 25, 5, 1, wb) == 0      # no owner kernel for 5x5
    assert lib.tvmi_roi_align_backward_overwrites(TVMI_F32, N, C, H, W, K, P, P, P * P, P, 1, wb // 2) == 0  # workspace too small
    ws = torch.empty(wb, dtype=torch.uint8, device=DEV)
    gin = torch.full((N, C, H, W), float("nan"), device=DEV)
    st = lib.tvmi_roi_align_backward(d_grad.data_ptr(), d_rois.data_ptr(), gin.data_ptr(), TVMI_F32, N, C, H, W, K, P, P, 1 / 8, 2, 0,
                                     C * P * P, P * P, P, 1, ws.data_ptr(), wb, _stream())
    assert st == 0, lib.tvmi_last_error()
    torch.cuda.synchronize()
    ref = O.roi_align_backward(grad.numpy(), rois.numpy(), 1 / 8, P, P, N, C, H, W, 2, False)
    np.testing.assert_allclose(gin.cpu().numpy(), ref, rtol=1e-4, atol=1e-4 * max(1.0, float(np.abs(ref).max())))
    # accumulate regime (no workspace): adds into a zero-filled buffer
    gin2 = torch.zeros(N, C, H, W, device=DEV)
    st = lib.tvmi_roi_align_backward(d_grad.data_ptr(), d_rois.data_ptr(), gin2.data_ptr(), TVMI_F32, N, C, H, W, K, P, P, 1 / 8, 2, 0,
                                     C * P * P, P * P, P, 1, None, 0, _stream())
    assert st == 0, lib.tvmi_last_error()
    torch.cuda.synchronize()
    np.testing.assert_allclose(gin2.cpu().numpy(), ref, rtol=1e-4, atol=1e-4 * max(1.0, float(np.abs(ref).max())))
