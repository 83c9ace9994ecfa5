"""Deformable convolution — host-side mirror of torchvision/ops/deform_conv.py
(deform_conv2d :14-107, DeformConv2d :110-204): same signature, the same dummy mask/bias
convention (:71-74), the same derivation of the group counts (:82-83) and the same error
for a malformed offset (:85-90)."""
import math
from typing import Optional, Tuple

import torch
from torch import Tensor, nn
from torch.nn import init
from torch.nn.parameter import Parameter

from ._loader import assert_has_ops
from .roi_ops import _pair


def deform_conv2d(input: Tensor, offset: Tensor, weight: Tensor, bias: Optional[Tensor] = None,
                  stride: Tuple[int, int] = (1, 1), padding: Tuple[int, int] = (0, 0),
                  dilation: Tuple[int, int] = (1, 1), mask: Optional[Tensor] = None) -> Tensor:
    """Deformable Convolution v2 (mask given) / v1 (mask None); Tensor[B, OC, oh, ow]."""
    assert_has_ops()
    out_channels = weight.shape[0]
    use_mask = mask is not None
    if mask is None:
        mask = torch.zeros((input.shape[0], 1), device=input.device, dtype=input.dtype)
    if bias is None:
        bias = torch.zeros(out_channels, device=input.device, dtype=input.dtype)
    stride_h, stride_w = _pair(stride)
    pad_h, pad_w = _pair(padding)
    dil_h, dil_w = _pair(dilation)
    kh, kw = weight.shape[-2:]
    n_in_channels = input.shape[1]
    n_offset_grps = offset.shape[1] // (2 * kh * kw)
    n_weight_grps = n_in_channels // weight.shape[1]
    if n_offset_grps == 0:
        raise RuntimeError(
            "the shape of the offset tensor at dimension 1 is not valid. It should "
            "be a multiple of 2 * weight.size[2] * weight.size[3].\n"
            f"Got offset.shape[1]={offset.shape[1]}, while 2 * weight.size[2] * weight.size[3]={2 * kh * kw}"
        )
    return torch.ops.torchvision.deform_conv2d(input, weight, offset, mask, bias, stride_h, stride_w, pad_h, pad_w,
                                               dil_h, dil_w, n_weight_grps, n_offset_grps, use_mask)


class DeformConv2d(nn.Module):
    """See :func:`deform_conv2d`; parameters and initialisation as torchvision.ops.DeformConv2d."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size, stride=1, padding=0, dilation=1,
                 groups: int = 1, bias: bool = True):
        super().__init__()
        if in_channels % groups != 0:
            raise ValueError("in_channels must be divisible by groups")
        if out_channels % groups != 0:
            raise ValueError("out_channels must be divisible by groups")
        self.in_channels, self.out_channels, self.groups = in_channels, out_channels, groups
        self.kernel_size, self.stride = _pair(kernel_size), _pair(stride)
        self.padding, self.dilation = _pair(padding), _pair(dilation)
        self.weight = Parameter(torch.empty(out_channels, in_channels // groups, *self.kernel_size))
        if bias:
            self.bias = Parameter(torch.empty(out_channels))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self) -> None:
        init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in, _ = init._calculate_fan_in_and_fan_out(self.weight)
            bound = 1 / math.sqrt(fan_in)
            init.uniform_(self.bias, -bound, bound)

    def forward(self, input: Tensor, offset: Tensor, mask: Optional[Tensor] = None) -> Tensor:
        return deform_conv2d(input, offset, self.weight, self.bias, stride=self.stride, padding=self.padding,
                             dilation=self.dilation, mask=mask)

    def __repr__(self) -> str:
        s = f"{self.__class__.__name__}({self.in_channels}, {self.out_channels}, kernel_size={self.kernel_size}"
        s += f", stride={self.stride}"
        s += f", padding={self.padding}" if self.padding != (0, 0) else ""
        s += f", dilation={self.dilation}" if self.dilation != (1, 1) else ""
        s += f", groups={self.groups}" if self.groups != 1 else ""
        s += ", bias=False" if self.bias is None else ""
        return s + ")"
