"""Deformable convolution v1 / v2 — the host-side entry of `torchvision::deform_conv2d` (deform_conv2d.hip).

Interface of torchvision/ops/deform_conv.py (function :14-107, module :110-204): argument order, the
"zero-sized placeholder" convention for an absent mask / bias (:71-74), group counts derived from the shapes
(:82-83) and the offset-shape error (:85-90).  The module is a `_ConvNd`: parameters, their initialisation,
`state_dict` keys and `repr` are the ones every torch convolution has, which is also what the reference's
hand-written module produces."""
from typing import Optional, Tuple, Union

import torch
from torch import Tensor
from torch.nn.modules.conv import _ConvNd

from ._loader import assert_has_ops
from .roi_ops import _pair

IntOrPair = Union[int, Tuple[int, int]]


def _placeholder(like: Tensor, *shape: int) -> Tensor:
    return like.new_zeros(shape)


def deform_conv2d(input: Tensor, offset: Tensor, weight: Tensor, bias: Optional[Tensor] = None, stride: IntOrPair = (1, 1),
                  padding: IntOrPair = (0, 0), dilation: IntOrPair = (1, 1), mask: Optional[Tensor] = None) -> Tensor:
    """input [B, C, H, W], offset [B, 2·G·kh·kw, oh, ow], weight [OC, C/groups, kh, kw], optional mask
    [B, G·kh·kw, oh, ow] (v2) → [B, OC, oh, ow].  `groups` = C ÷ weight.shape[1], the offset groups G follow from the
    offset's channel count."""
    assert_has_ops()
    taps = weight.shape[-2] * weight.shape[-1]
    offset_groups, weight_groups = offset.shape[1] // (2 * taps), input.shape[1] // weight.shape[1]
    if offset_groups == 0:
        raise RuntimeError("the shape of the offset tensor at dimension 1 is not valid. It should be a multiple of "
                           "2 * weight.size[2] * weight.size[3].\n"
                           f"Got offset.shape[1]={offset.shape[1]}, while 2 * weight.size[2] * weight.size[3]={2 * taps}")
    (sh, sw), (ph, pw), (dh, dw) = _pair(stride), _pair(padding), _pair(dilation)
    return torch.ops.torchvision.deform_conv2d(
        input, weight, offset,
        _placeholder(input, input.shape[0], 1) if mask is None else mask,
        _placeholder(input, weight.shape[0]) if bias is None else bias,
        sh, sw, ph, pw, dh, dw, weight_groups, offset_groups, mask is not None)


class DeformConv2d(_ConvNd):
    """Module form of :func:`deform_conv2d` (torchvision.ops.DeformConv2d); offsets (and the v2 mask) are inputs of
    `forward`, produced by a sibling convolution in the caller's network."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: IntOrPair, stride: IntOrPair = 1, padding: IntOrPair = 0,
                 dilation: IntOrPair = 1, groups: int = 1, bias: bool = True):
        super().__init__(in_channels, out_channels, _pair(kernel_size), _pair(stride), _pair(padding), _pair(dilation),
                         False, (0, 0), groups, bias, "zeros")

    def forward(self, input: Tensor, offset: Tensor, mask: Optional[Tensor] = None) -> Tensor:
        return deform_conv2d(input, offset, self.weight, self.bias, self.stride, self.padding, self.dilation, mask)
