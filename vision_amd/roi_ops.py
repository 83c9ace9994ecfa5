"""RoI pooling operators — host-side mirror of torchvision/ops/{roi_align,roi_pool,
ps_roi_align,ps_roi_pool}.py (function + nn.Module pairs, same argument meaning) and of
ops/_utils.py:18-37 (box-list -> [K,5] conversion, shape checks)."""
from typing import List, Sequence, Tuple, Union

import torch
from torch import Tensor, nn

from ._loader import assert_has_ops

BoxesArg = Union[Tensor, Sequence[Tensor]]


def _pair(v) -> Tuple[int, int]:
    if isinstance(v, int):
        return (v, v)
    v = tuple(v)
    return (v[0], v[0]) if len(v) == 1 else (v[0], v[1])


def check_roi_boxes_shape(boxes: BoxesArg):
    if isinstance(boxes, (list, tuple)):
        for t in boxes:
            torch._assert(t.size(1) == 4,
                          "The shape of the tensor in the boxes list is not correct as List[Tensor[L, 4]]")
    elif isinstance(boxes, Tensor):
        torch._assert(boxes.size(1) == 5, "The boxes tensor shape is not correct as Tensor[K, 5]")
    else:
        torch._assert(False, "boxes is expected to be a Tensor[L, 5] or a List[Tensor[K, 4]]")


def convert_boxes_to_roi_format(boxes: Sequence[Tensor]) -> Tensor:
    """List of per-image [L,4] boxes -> [K,5] with the image index in column 0 (ops/_utils.py:18-25).
    Device tensors: one launch (`tvmi::boxes_to_rois`)."""
    boxes = list(boxes)
    if (boxes and boxes[0].is_cuda and len(boxes) <= 64 and boxes[0].is_floating_point()
            and all(b.dtype == boxes[0].dtype and not b.requires_grad for b in boxes)):
        return torch.ops.tvmi.boxes_to_rois(boxes)
    ids = [torch.full_like(b[:, :1], i) for i, b in enumerate(boxes)]
    return torch.cat([torch.cat(ids, dim=0), torch.cat(list(boxes), dim=0)], dim=1)


def _rois(boxes: BoxesArg) -> Tensor:
    check_roi_boxes_shape(boxes)
    return boxes if isinstance(boxes, Tensor) else convert_boxes_to_roi_format(boxes)


def roi_align(input: Tensor, boxes: BoxesArg, output_size, spatial_scale: float = 1.0,
              sampling_ratio: int = -1, aligned: bool = False) -> Tensor:
    """Mask R-CNN RoIAlign, Tensor[K, C, oh, ow] (torchvision.ops.roi_align,
    ops/roi_align.py:204-285)."""
    assert_has_ops()
    oh, ow = _pair(output_size)
    rois = _rois(boxes)
    if input.is_quantized:      # ops/roi_align.py:251-274: per-tensor quantized input and rois -> qroi_align on the integer data
        if not rois.is_quantized:
            raise ValueError("If input is quantized, rois must also be quantized.")
        out_int = torch.ops.torchvision.qroi_align(input.int_repr(), rois.int_repr(), input.q_scale(), input.q_zero_point(),
                                                   rois.q_scale(), rois.q_zero_point(), spatial_scale, oh, ow, sampling_ratio, aligned)
        return torch._make_per_tensor_quantized_tensor(out_int, scale=input.q_scale(), zero_point=input.q_zero_point())
    return torch.ops.torchvision.roi_align(input, rois, spatial_scale, oh, ow, sampling_ratio, aligned)


def roi_pool(input: Tensor, boxes: BoxesArg, output_size, spatial_scale: float = 1.0) -> Tensor:
    """Fast R-CNN RoIPool (ops/roi_pool.py:14-53)."""
    assert_has_ops()
    oh, ow = _pair(output_size)
    return torch.ops.torchvision.roi_pool(input, _rois(boxes), spatial_scale, oh, ow)[0]


def ps_roi_align(input: Tensor, boxes: BoxesArg, output_size, spatial_scale: float = 1.0,
                 sampling_ratio: int = -1) -> Tensor:
    """Position-sensitive RoIAlign (ops/ps_roi_align.py:14-60)."""
    assert_has_ops()
    oh, ow = _pair(output_size)
    return torch.ops.torchvision.ps_roi_align(input, _rois(boxes), spatial_scale, oh, ow, sampling_ratio)[0]


def ps_roi_pool(input: Tensor, boxes: BoxesArg, output_size, spatial_scale: float = 1.0) -> Tensor:
    """Position-sensitive RoIPool (ops/ps_roi_pool.py:14-54)."""
    assert_has_ops()
    oh, ow = _pair(output_size)
    return torch.ops.torchvision.ps_roi_pool(input, _rois(boxes), spatial_scale, oh, ow)[0]


class RoIAlign(nn.Module):
    def __init__(self, output_size, spatial_scale: float, sampling_ratio: int, aligned: bool = False):
        super().__init__()
        self.output_size, self.spatial_scale = output_size, spatial_scale
        self.sampling_ratio, self.aligned = sampling_ratio, aligned

    def forward(self, input: Tensor, rois: BoxesArg) -> Tensor:
        return roi_align(input, rois, self.output_size, self.spatial_scale, self.sampling_ratio, self.aligned)

    def __repr__(self) -> str:
        return (f"{self.__class__.__name__}(output_size={self.output_size}, spatial_scale={self.spatial_scale}, "
                f"sampling_ratio={self.sampling_ratio}, aligned={self.aligned})")


class RoIPool(nn.Module):
    def __init__(self, output_size, spatial_scale: float):
        super().__init__()
        self.output_size, self.spatial_scale = output_size, spatial_scale

    def forward(self, input: Tensor, rois: BoxesArg) -> Tensor:
        return roi_pool(input, rois, self.output_size, self.spatial_scale)

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}(output_size={self.output_size}, spatial_scale={self.spatial_scale})"


class PSRoIAlign(nn.Module):
    def __init__(self, output_size, spatial_scale: float, sampling_ratio: int):
        super().__init__()
        self.output_size, self.spatial_scale, self.sampling_ratio = output_size, spatial_scale, sampling_ratio

    def forward(self, input: Tensor, rois: BoxesArg) -> Tensor:
        return ps_roi_align(input, rois, self.output_size, self.spatial_scale, self.sampling_ratio)

    def __repr__(self) -> str:
        return (f"{self.__class__.__name__}(output_size={self.output_size}, spatial_scale={self.spatial_scale}, "
                f"sampling_ratio={self.sampling_ratio})")


class PSRoIPool(nn.Module):
    def __init__(self, output_size, spatial_scale: float):
        super().__init__()
        self.output_size, self.spatial_scale = output_size, spatial_scale

    def forward(self, input: Tensor, rois: BoxesArg) -> Tensor:
        return ps_roi_pool(input, rois, self.output_size, self.spatial_scale)

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}(output_size={self.output_size}, spatial_scale={self.spatial_scale})"
