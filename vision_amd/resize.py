"""Resize / interpolation — host-side mirror of the reference's resize call path:

  torchvision.transforms.v2.functional.resize_image   (_geometry.py:283-362)
  torchvision.transforms._functional_tensor.resize     (_functional_tensor.py:441-474)
  torch.nn.functional.interpolate (4-D, the modes those wrappers use)

`interpolate` has F.interpolate's argument meaning for 4-D NCHW inputs and the modes
nearest / nearest-exact / bilinear / bicubic (+ antialias); `resize` has resize_image's
meaning (size int -> shorter edge, max_size, uint8 handled as float32 + round + clamp exactly
like the reference does on GPU tensors, _geometry.py:316-360).  CUDA tensors run on
`tvmi::interpolate2d`; there is no fallback to ATen for the forward.  Inputs that require grad get
`tvmi::interpolate2d_backward` as the gradient (gather kernels, deterministic; float64 keeps ATen's); the nearest modes
take any element type (uint8 images / masks stay uint8, as in the reference).
"""
import math
from typing import List, Optional, Sequence, Union

import torch
from torch import Tensor

from ._loader import assert_has_ops

_MODES = {"nearest": 0, "nearest-exact": 1, "bilinear": 2, "bicubic": 3}


def interpolate(input: Tensor, size=None, scale_factor=None, mode: str = "nearest",
                align_corners: Optional[bool] = None, recompute_scale_factor: Optional[bool] = None,
                antialias: bool = False) -> Tensor:
    """torch.nn.functional.interpolate for 4-D inputs on the gfx950 kernels."""
    assert_has_ops()
    if input.dim() != 4:
        raise NotImplementedError("vision_amd.interpolate handles 4-D [N, C, H, W] inputs only")
    if mode not in _MODES:
        raise NotImplementedError(f"interpolation mode {mode!r} is not on the resize path")
    if mode in ("nearest", "nearest-exact"):
        if align_corners is not None:
            raise ValueError("align_corners option can only be set with the interpolating modes: "
                             "linear | bilinear | bicubic | trilinear")
        align = False
    else:
        align = bool(align_corners) if align_corners is not None else False
    if antialias and mode not in ("bilinear", "bicubic"):
        raise ValueError("Anti-alias option is restricted to bilinear and bicubic modes and requires a 3-D, 4-D "
                         "or 5-D input tensor")
    if (size is None) == (scale_factor is None):
        raise ValueError("only one of size or scale_factor should be defined" if size is not None
                         else "either size or scale_factor should be defined")
    ih, iw = input.shape[-2:]
    scale_h = scale_w = -1.0
    if size is not None:
        oh, ow = (size, size) if isinstance(size, int) else tuple(int(s) for s in size)
    else:
        sf = (scale_factor, scale_factor) if isinstance(scale_factor, (int, float)) else tuple(scale_factor)
        # torch/nn/functional.py: output size = floor(input size * scale factor)
        oh, ow = int(math.floor(float(ih) * sf[0])), int(math.floor(float(iw) * sf[1]))
        if not recompute_scale_factor:
            scale_h, scale_w = float(sf[0]), float(sf[1])
    if input.requires_grad and torch.is_grad_enabled():
        # forward and backward on the resize kernels of this library
        return _Interpolate2d.apply(input, oh, ow, mode, align, bool(antialias), scale_h, scale_w)
    return torch.ops.tvmi.interpolate2d(input, oh, ow, _MODES[mode], align, bool(antialias), scale_h, scale_w)


class _Interpolate2d(torch.autograd.Function):
    """tvmi::interpolate2d with tvmi::interpolate2d_backward as its gradient: the gather kernels of resize.hip (a lane owns an
    input pixel and sums its output range in a fixed order — bit-reproducible, where ATen's bilinear / bicubic / anti-aliased
    backward kernels scatter with atomicAdd).  float64 gradients keep ATen's `upsample_*_backward` kernels."""

    @staticmethod
    def forward(ctx, input, oh, ow, mode, align, antialias, scale_h, scale_w):
        ctx.cfg = (tuple(input.shape), oh, ow, mode, align, antialias, scale_h, scale_w)
        return torch.ops.tvmi.interpolate2d(input, oh, ow, _MODES[mode], align, antialias, scale_h, scale_w)

    @staticmethod
    def backward(ctx, grad):
        shape, oh, ow, mode, align, antialias, scale_h, scale_w = ctx.cfg
        # create_graph=True (grad mode on inside backward): ATen's `upsample_*_backward` ops carry a derivative formula,
        # tvmi::interpolate2d_backward does not — double backward keeps working through them.  A channels_last
        # grad is read through a contiguous copy and the gradient comes back NCHW-contiguous.
        if grad.dtype in (torch.float32, torch.float16, torch.bfloat16) and not (torch.is_grad_enabled() and grad.requires_grad):
            gi = torch.ops.tvmi.interpolate2d_backward(grad, shape[2], shape[3], _MODES[mode], align, antialias, scale_h, scale_w)
            return gi, None, None, None, None, None, None, None
        sh = None if scale_h <= 0 else scale_h
        sw = None if scale_w <= 0 else scale_w
        a, g = torch.ops.aten, grad.contiguous()
        if mode == "nearest":
            gi = a.upsample_nearest2d_backward(g, [oh, ow], list(shape), sh, sw)
        elif mode == "nearest-exact":
            gi = a._upsample_nearest_exact2d_backward(g, [oh, ow], list(shape), sh, sw)
        elif mode == "bilinear":
            op = a._upsample_bilinear2d_aa_backward if antialias else a.upsample_bilinear2d_backward
            gi = op(g, [oh, ow], list(shape), align, sh, sw)
        else:
            op = a._upsample_bicubic2d_aa_backward if antialias else a.upsample_bicubic2d_backward
            gi = op(g, [oh, ow], list(shape), align, sh, sw)
        return gi, None, None, None, None, None, None, None


def _compute_resized_output_size(canvas_size, size: Optional[Sequence[int]], max_size: Optional[int] = None) -> List[int]:
    # torchvision/transforms/v2/functional/_geometry.py:191-246 / transforms/functional.py
    h, w = canvas_size
    if size is None:
        if not isinstance(max_size, int):
            raise ValueError(f"max_size must be an integer when size is None, but got {max_size} instead.")
        short, long = (w, h) if w <= h else (h, w)
        new_long, new_short = max_size, int(max_size * short / long)
        return [new_long, new_short] if w <= h else [new_short, new_long]
    if isinstance(size, int):
        size = [size]
    if len(size) == 1:
        short, long = (w, h) if w <= h else (h, w)
        requested = size[0]
        new_short, new_long = requested, int(requested * long / short)
        if max_size is not None:
            if max_size <= requested:
                raise ValueError(f"max_size = {max_size} must be strictly greater than the requested "
                                 f"size for the smaller edge size = {size}")
            if new_long > max_size:
                new_short, new_long = int(max_size * new_short / new_long), max_size
        return [new_long, new_short] if w <= h else [new_short, new_long]
    return [int(size[0]), int(size[1])]


def resize(image: Tensor, size: Optional[Union[int, Sequence[int]]], interpolation: str = "bilinear",
           max_size: Optional[int] = None, antialias: Optional[bool] = True) -> Tensor:
    """resize_image of the reference for tensors [..., C, H, W] (bilinear / bicubic / nearest /
    nearest-exact given as strings, like InterpolationMode.value)."""
    antialias = bool(antialias) and interpolation in ("bilinear", "bicubic")
    shape = image.shape
    old_h, old_w = shape[-2:]
    new_h, new_w = _compute_resized_output_size((old_h, old_w), size=size, max_size=max_size)
    if (new_h, new_w) == (old_h, old_w):
        return image
    x = image.reshape(-1, shape[-3] if image.dim() >= 3 else 1, old_h, old_w)
    dtype = x.dtype
    # _geometry.py:316-342: uint8 stays uint8 in the nearest modes (the op is a copy), other integer types go through float32
    need_cast = not (x.is_floating_point() or (dtype == torch.uint8 and interpolation in ("nearest", "nearest-exact")))
    if need_cast:
        x = x.to(torch.float32)
    align = False if interpolation in ("bilinear", "bicubic") else None
    x = interpolate(x, size=[new_h, new_w], mode=interpolation, align_corners=align, antialias=antialias)
    if need_cast:
        if interpolation == "bicubic" and dtype == torch.uint8:
            x = x.clamp_(min=0, max=255)
        if dtype in (torch.uint8, torch.int8, torch.int16, torch.int32, torch.int64):
            x = x.round_()
        x = x.to(dtype)
    return x.reshape(shape[:-2] + (new_h, new_w))
