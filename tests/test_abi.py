"""The drop-in boundary: the C-ABI library loads and exports every symbol include/tvmi.h
declares; the dispatcher glue defines the reference's schemas verbatim.  (No GPU here: the compute calls straight
through the C ABI are in tests/test_gpu_abi.py.)"""
import ctypes
import os
import re

import pytest
import torch

from helpers import ROOT

HEADER = os.path.join(ROOT, "include", "tvmi.h")

REFERENCE_SCHEMAS = {
    "nms": "torchvision::nms(Tensor dets, Tensor scores, float iou_threshold) -> Tensor",
    "roi_align": "torchvision::roi_align(Tensor input, Tensor rois, float spatial_scale, SymInt pooled_height, SymInt pooled_width, int sampling_ratio, bool aligned) -> Tensor",
    "_roi_align_backward": "torchvision::_roi_align_backward(Tensor grad, Tensor rois, float spatial_scale, SymInt pooled_height, SymInt pooled_width, SymInt batch_size, SymInt channels, SymInt height, SymInt width, int sampling_ratio, bool aligned) -> Tensor",
    "roi_pool": "torchvision::roi_pool(Tensor input, Tensor rois, float spatial_scale, SymInt pooled_height, SymInt pooled_width) -> (Tensor, Tensor)",
    "_roi_pool_backward": "torchvision::_roi_pool_backward(Tensor grad, Tensor rois, Tensor argmax, float spatial_scale, SymInt pooled_height, SymInt pooled_width, SymInt batch_size, SymInt channels, SymInt height, SymInt width) -> Tensor",
    "ps_roi_align": "torchvision::ps_roi_align(Tensor input, Tensor rois, float spatial_scale, SymInt pooled_height, SymInt pooled_width, int sampling_ratio) -> (Tensor, Tensor)",
    "_ps_roi_align_backward": "torchvision::_ps_roi_align_backward(Tensor grad, Tensor rois, Tensor channel_mapping, float spatial_scale, SymInt pooled_height, SymInt pooled_width, int sampling_ratio, SymInt batch_size, SymInt channels, SymInt height, SymInt width) -> Tensor",
    "ps_roi_pool": "torchvision::ps_roi_pool(Tensor input, Tensor rois, float spatial_scale, SymInt pooled_height, SymInt pooled_width) -> (Tensor, Tensor)",
    "_ps_roi_pool_backward": "torchvision::_ps_roi_pool_backward(Tensor grad, Tensor rois, Tensor channel_mapping, float spatial_scale, SymInt pooled_height, SymInt pooled_width, SymInt batch_size, SymInt channels, SymInt height, SymInt width) -> Tensor",
    "deform_conv2d": "torchvision::deform_conv2d(Tensor input, Tensor weight, Tensor offset, Tensor mask, Tensor bias, SymInt stride_h, SymInt stride_w, SymInt pad_h, SymInt pad_w, SymInt dilation_h, SymInt dilation_w, SymInt groups, SymInt offset_groups, bool use_mask) -> Tensor",
    "_deform_conv2d_backward": "torchvision::_deform_conv2d_backward(Tensor grad, Tensor input, Tensor weight, Tensor offset, Tensor mask, Tensor bias, SymInt stride_h, SymInt stride_w, SymInt pad_h, SymInt pad_w, SymInt dilation_h, SymInt dilation_w, SymInt groups, SymInt offset_groups, bool use_mask) -> (Tensor, Tensor, Tensor, Tensor, Tensor)",
    "box_iou_rotated": "torchvision::box_iou_rotated(Tensor boxes1, Tensor boxes2) -> Tensor",
    "qnms": "torchvision::qnms(Tensor dets, Tensor scores, float iou_threshold) -> Tensor",
    "qroi_align": "torchvision::qroi_align(Tensor input, Tensor rois, float input_scale, int input_zero_point, float rois_scale, int rois_zero_point, float spatial_scale, SymInt pooled_height, SymInt pooled_width, int sampling_ratio, bool aligned) -> Tensor",
    "_cuda_version": "torchvision::_cuda_version() -> int _0",  # inferred from the C++ signature, as in vision.cpp:31
}


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tvmi_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_whole_path():
    syms = declared_symbols()
    for needed in ("tvmi_nms", "tvmi_roi_align_forward", "tvmi_roi_align_backward", "tvmi_roi_pool_forward",
                   "tvmi_ps_roi_align_forward", "tvmi_ps_roi_pool_forward", "tvmi_deform_conv2d_forward",
                   "tvmi_box_iou_rotated", "tvmi_upsample_bilinear2d", "tvmi_upsample_bicubic2d"):
        assert needed in syms


def test_kernel_library_exports_every_declared_symbol():
    import vision_amd

    lib = vision_amd._loader.kernels()
    for sym in declared_symbols():
        assert hasattr(lib, sym), f"libtvmi_kernels.so does not export {sym}"
    lib.tvmi_version.restype = ctypes.c_int
    header = open(os.path.join(ROOT, 'include', 'tvmi.h')).read()
    import re as _re
    assert lib.tvmi_version() == int(_re.search(r'#define TVMI_ABI_VERSION (\d+)', header).group(1)) == 307   # ABI bumped with the changed entries (round 6: tvmi_nms_step, the one-launch detector step)
    lib.tvmi_arch.restype = ctypes.c_char_p
    assert lib.tvmi_arch() == b"gfx950"


def test_kernel_library_is_pure_c_abi():
    # the kernels library must not depend on torch (only the HIP runtime and libc/libstdc++)
    import subprocess

    import vision_amd

    out = subprocess.check_output(["readelf", "-d", vision_amd._loader.KERNELS_SO], text=True)
    needed = re.findall(r"NEEDED.*\[(.*?)\]", out)
    assert any("amdhip64" in n for n in needed)
    assert not any("torch" in n or "c10" in n for n in needed), needed


def test_stable_glue_is_stable_abi_only():
    """tvmi_torch_stable.so (the `_C_stable` role: CUDA kernels of nms and box_iou_rotated, the two ops the reference keeps
    on the stable ABI — cuda/nms_kernel.cu:262, cuda/box_iou_rotated_kernel.cu:192) must reach torch only through the C
    shim: no ATen / c10 C++ symbol among its undefined symbols, no libc10 among its dependencies."""
    import subprocess

    import vision_amd

    so = vision_amd._loader.STABLE_SHIM_SO
    undefined = subprocess.check_output(["nm", "-D", "--undefined-only", so], text=True).split("\n")
    cxx_torch = [u for u in undefined if "_ZN2at" in u or "_ZN3c10" in u or "_ZNK2at" in u or "_ZNK3c10" in u or "_ZN5torch" in u]
    assert not cxx_torch, cxx_torch[:5]
    assert any("torch_call_dispatcher" in u or "aoti_torch_" in u for u in undefined)
    needed = re.findall(r"NEEDED.*\[(.*?)\]", subprocess.check_output(["readelf", "-d", so], text=True))
    assert not any("c10" in n for n in needed), needed
    # and the classic glue no longer carries those two kernels: they are registered exactly once
    assert torch._C._dispatch_has_kernel_for_dispatch_key("torchvision::nms", "CUDA")
    assert torch._C._dispatch_has_kernel_for_dispatch_key("torchvision::box_iou_rotated", "CUDA")


def test_schemas_are_the_reference_strings(tv):
    for name, schema in REFERENCE_SCHEMAS.items():
        op = getattr(tv, name).default
        assert str(op._schema) == schema, name


def test_cuda_key_has_kernels_for_every_op(tv):
    for name in REFERENCE_SCHEMAS:
        if name in ("qnms", "qroi_align", "_cuda_version"):
            continue
        assert torch._C._dispatch_has_kernel_for_dispatch_key(f"torchvision::{name}", "CUDA"), name
    assert tv._cuda_version() == -1
    assert torch.ops.tvmi.abi_version() == 307


def test_fake_kernels_give_reference_shapes(tv):
    from torch._subclasses.fake_tensor import FakeTensorMode

    with FakeTensorMode():
        x = torch.empty(2, 50, 10, 12)
        rois = torch.empty(7, 5)
        assert tv.roi_align(x, rois, 1.0, 5, 4, 2, False).shape == (7, 50, 5, 4)
        o, a = tv.roi_pool(x, rois, 1.0, 5, 5)
        assert o.shape == (7, 50, 5, 5) and a.dtype == torch.int32
        o, m = tv.ps_roi_align(x, rois, 1.0, 5, 5, 2)
        assert o.shape == (7, 2, 5, 5) and m.dtype == torch.int32
        o, m = tv.ps_roi_pool(x, rois, 1.0, 5, 5)
        assert o.shape == (7, 2, 5, 5)
        y = tv.deform_conv2d(torch.empty(2, 6, 9, 9), torch.empty(4, 3, 3, 3), torch.empty(2, 18, 7, 7),
                             torch.empty(2, 9, 7, 7), torch.empty(4), 1, 1, 0, 0, 1, 1, 2, 1, True)
        assert y.shape == (2, 4, 7, 7)
        assert tv.box_iou_rotated(torch.empty(3, 5), torch.empty(4, 5)).dtype == torch.float32


def test_missing_extension_fails_loudly(tmp_path, monkeypatch):
    import vision_amd._loader as L

    monkeypatch.setattr(L, "KERNELS_SO", str(tmp_path / "nope.so"))
    monkeypatch.setitem(L._state, "loaded", False)
    with pytest.raises(L.ExtensionMissing):
        L.load()
    monkeypatch.setitem(L._state, "loaded", True)


def test_deform_conv2d_backward_workspace_plan_is_host_side_and_covers_its_buffers():
    """tvmi_deform_conv2d_backward_workspace_bytes is a pure host query (no device needed): it must cover what the route of a
    shape places in the workspace (deform_conv2d_bwd.hip, bwd_plan) and be 0 where nothing is needed."""
    import vision_amd

    lib = vision_amd._loader.kernels()
    f = lib.tvmi_deform_conv2d_backward_workspace_bytes
    f.restype = ctypes.c_size_t
    f.argtypes = [ctypes.c_int] + [ctypes.c_int64] * 15
    header = open(HEADER).read()
    dt = {k: int(re.search(rf"\b{k}\s*=\s*(\d+)", header).group(1)) for k in ("TVMI_F32", "TVMI_F64", "TVMI_F16", "TVMI_BF16")}
    B, C, H, W, OC, k = 2, 256, 100, 136, 256, 3
    n_in, n_out, KK = B * C * H * W, B * OC * H * W, k * k
    # groups = 1, fp32: re-laid-out weights + [tap][oc][ic] sums + channels-last input copy + channels-last grad_input sums
    g1 = f(dt["TVMI_F32"], B, C, H, W, OC, k, k, 1, 1, 1, 1, 1, 1, 1, 1)
    assert g1 >= 4 * (KK * OC * C) * 2 + 4 * n_in * 2
    # 16-bit: additionally fp32 sums of grad_offset / grad_mask (and grad_input for the routes that want it planar)
    g1h = f(dt["TVMI_BF16"], B, C, H, W, OC, k, k, 1, 1, 1, 1, 1, 1, 1, 1)
    assert g1h >= 4 * (KK * OC * C) * 2 + 2 * n_in + 4 * n_in + 4 * B * 3 * KK * H * W
    # depthwise: channels-last copies of input and grad_out, channels-last sums, fp32 weight-gradient sums
    dw = f(dt["TVMI_F32"], B, C, H, W, OC, k, k, 1, 1, 1, 1, 1, 1, C, 1)
    assert dw >= 4 * n_in + 4 * n_out + 4 * n_in + 4 * OC * KK
    # fp64 runs the direct kernels in place; nonsense shapes ask for nothing
    assert f(dt["TVMI_F64"], B, C, H, W, OC, k, k, 1, 1, 1, 1, 1, 1, 1, 1) == 0
    assert f(dt["TVMI_F32"], 0, C, H, W, OC, k, k, 1, 1, 1, 1, 1, 1, 1, 1) == 0
    assert f(dt["TVMI_F32"], B, C, H, W, OC, k, k, 1, 1, 1, 1, 1, 1, 3, 1) == 0      # C % groups != 0
    # the owner form's channels-last buffers (2 x 4 bytes per input element) are asked for only where that form can
    # run — a dilation-7 window does not fit its LDS, a 1 x 9 kernel is not its shape
    assert f(dt["TVMI_F32"], B, C, H, W, OC, k, k, 1, 1, 7, 7, 7, 7, 1, 1) <= g1 - 8 * n_in
    assert f(dt["TVMI_F32"], B, C, H, W, OC, 1, 9, 1, 1, 0, 4, 1, 1, 1, 1) <= g1 - 8 * n_in


def test_resize_backward_and_channels_last_entries_host_side():
    """The workspace queries of the round-5 resize entries are pure host arithmetic, and the launchers check their arguments
    before anything touches a device: table sizes per mode (1 / 2 / 4 taps; the anti-aliased modes by scale), the ranges of the
    backward on top, and the refusals (unknown mode, anti-aliasing with a nearest mode, float64, missing workspace)."""
    import vision_amd

    lib = vision_amd._loader.kernels()
    header = open(HEADER).read()
    dt = {k: int(re.search(rf"\b{k}\s*=\s*(\d+)", header).group(1)) for k in ("TVMI_F32", "TVMI_F64", "TVMI_F16", "TVMI_BF16")}
    wb, wn = lib.tvmi_upsample2d_backward_workspace_bytes, lib.tvmi_upsample2d_nhwc_workspace_bytes
    for f in (wb, wn):
        f.restype = ctypes.c_size_t
        f.argtypes = [ctypes.c_int, ctypes.c_int] + [ctypes.c_int64] * 4 + [ctypes.c_int, ctypes.c_double, ctypes.c_double]
    IH, IW, OH, OW = 100, 168, 200, 336
    for mode, taps in ((0, 1), (1, 1), (2, 2), (3, 4)):
        tables = 4 * (OH * (taps + 2) + OW * (taps + 2))
        assert wn(mode, 0, IH, IW, OH, OW, 0, -1.0, -1.0) == tables                      # [first, taps, weights] per output index
        assert wb(mode, 0, IH, IW, OH, OW, 0, -1.0, -1.0) == tables + 8 * (IH + IW)       # + one [lo, hi) range per input index
    # anti-aliased down-scale by 2.5: support 2.5 (bilinear) -> 7 taps per output
    aa = wn(2, 1, 250, 250, 100, 100, 0, -1.0, -1.0)
    assert aa == 4 * 2 * 100 * (7 + 2)
    assert wb(2, 0, 0, IW, OH, OW, 0, -1.0, -1.0) == 0 and wb(7, 0, IH, IW, OH, OW, 0, -1.0, -1.0) == 0
    bwd = lib.tvmi_upsample2d_backward
    bwd.restype = ctypes.c_int
    bwd.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int] + [ctypes.c_int64] * 5 + \
                   [ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    lib.tvmi_last_error.restype = ctypes.c_char_p
    fake = ctypes.c_void_p(4096)    # never dereferenced: every call below is refused first
    assert bwd(fake, fake, dt["TVMI_F32"], 0, 0, 0, IH, IW, OH, OW, 0, -1.0, -1.0, None, 0, None) == 0     # empty problem
    assert bwd(fake, fake, dt["TVMI_F32"], 9, 0, 4, IH, IW, OH, OW, 0, -1.0, -1.0, fake, 1 << 20, None) != 0
    assert b"mode" in lib.tvmi_last_error()
    assert bwd(fake, fake, dt["TVMI_F32"], 0, 1, 4, IH, IW, OH, OW, 0, -1.0, -1.0, fake, 1 << 20, None) != 0
    assert b"anti-aliasing" in lib.tvmi_last_error()
    assert bwd(fake, fake, dt["TVMI_F64"], 2, 0, 4, IH, IW, OH, OW, 0, -1.0, -1.0, fake, 1 << 20, None) != 0
    assert b"float32" in lib.tvmi_last_error()
    assert bwd(fake, fake, dt["TVMI_F32"], 2, 0, 4, IH, IW, OH, OW, 0, -1.0, -1.0, None, 0, None) != 0
    assert b"workspace" in lib.tvmi_last_error()
    nn = lib.tvmi_upsample_nearest2d_any
    nn.restype = ctypes.c_int
    nn.argtypes = [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int64] * 6 + [ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_void_p]
    assert nn(fake, fake, 3, 4, IH, IW, OH, OW, 0, -1.0, -1.0, None) != 0 and b"element size" in lib.tvmi_last_error()
    # stream helpers: argument checks only (no device here)
    assert lib.tvmi_stream_event_scope(ctypes.c_int(5)) != 0 and lib.tvmi_stream_event_scope(ctypes.c_int(1)) == 0


def test_options_round_trip_host_side():
    """tvmi_set_option / tvmi_get_option are host-only switches: every name include/tvmi.h lists for the RoIAlign forward reads back
    what was set (no GPU needed), the tuning value is clamped to its documented range, unknown names are refused with an error text."""
    import ctypes

    import vision_amd

    lib = vision_amd._loader.kernels()
    lib.tvmi_set_option.argtypes = [ctypes.c_char_p, ctypes.c_int64]
    lib.tvmi_get_option.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_int64)]
    lib.tvmi_last_error.restype = ctypes.c_char_p

    def get(name):
        v = ctypes.c_int64(-99)
        assert lib.tvmi_get_option(name, ctypes.byref(v)) == 0, name
        return v.value

    header = open(HEADER).read()
    for name, default in ((b"roi_align.pin_chunks", 1), (b"roi_align.order", 1), (b"roi_align.inline_mop", 1), (b"roi_align.carry_step", 1),
                          (b"roi_align.fold_order", 1)):
        assert name.decode() in header or name in (b"roi_align.pin_chunks", b"roi_align.order"), name
        assert get(name) == default, name
        assert lib.tvmi_set_option(name, 0) == 0 and get(name) == 0
        assert lib.tvmi_set_option(name, default) == 0 and get(name) == default
    assert get(b"roi_align.fold_first_round_pct") == 100
    assert lib.tvmi_set_option(b"roi_align.fold_first_round_pct", 100000) == 0 and get(b"roi_align.fold_first_round_pct") == 400
    assert lib.tvmi_set_option(b"roi_align.fold_first_round_pct", 100) == 0
    assert lib.tvmi_set_option(b"roi_align.no_such_switch", 1) != 0
    assert b"unknown option" in lib.tvmi_last_error()
