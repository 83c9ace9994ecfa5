"""MultiScaleRoIAlign — host-side mirror of torchvision/ops/poolers.py (LevelMapper :47-84,
scale inference :97-134, _multiscale_roi_align :147-227, module :230-321)."""
from typing import Dict, List, Optional, Tuple, Union

import torch
from torch import Tensor, nn

from .boxes import box_area
from .roi_ops import roi_align


class LevelMapper:
    """FPN eq. (1): level = floor(k0 + log2(sqrt(area) / s0) + eps), clamped to [k_min, k_max]."""

    def __init__(self, k_min: int, k_max: int, canonical_scale: int = 224, canonical_level: int = 4,
                 eps: float = 1e-6):
        self.k_min, self.k_max = k_min, k_max
        self.s0, self.lvl0, self.eps = canonical_scale, canonical_level, eps

    def __call__(self, boxlists: List[Tensor]) -> Tensor:
        s = torch.sqrt(torch.cat([box_area(b) for b in boxlists]))
        target = torch.floor(self.lvl0 + torch.log2(s / self.s0) + torch.tensor(self.eps, dtype=s.dtype))
        target = torch.clamp(target, min=self.k_min, max=self.k_max)
        return (target.to(torch.int64) - self.k_min).to(torch.int64)


def _convert_to_roi_format(boxes: List[Tensor]) -> Tensor:
    """ops/_utils.py:18-25.  On device tensors this is ONE launch (`tvmi::boxes_to_rois`) instead of the
    reference's cat + full_like per image + two more cats."""
    if len(boxes) and boxes[0].is_cuda and len(boxes) <= 64 and boxes[0].is_floating_point() and \
            all(b.dtype == boxes[0].dtype and not b.requires_grad for b in boxes):
        return torch.ops.tvmi.boxes_to_rois(list(boxes))
    cat = torch.cat(boxes, dim=0)
    ids = torch.cat([torch.full_like(b[:, :1], i, dtype=cat.dtype, device=cat.device) for i, b in enumerate(boxes)],
                    dim=0)
    return torch.cat([ids, cat], dim=1)


def _infer_scale(feature: Tensor, original_size: Tuple[int, int]) -> float:
    # the scale is assumed to be 2 ** (-k) with integer k (poolers.py:97-106)
    scales = []
    for s1, s2 in zip(feature.shape[-2:], original_size):
        scales.append(2 ** float(torch.tensor(float(s1) / float(s2)).log2().round()))
    return scales[0]


def _setup_scales(features: List[Tensor], image_shapes: List[Tuple[int, int]], canonical_scale: int,
                  canonical_level: int):
    if not image_shapes:
        raise ValueError("images list should not be empty")
    max_x = max(s[0] for s in image_shapes)
    max_y = max(s[1] for s in image_shapes)
    scales = [_infer_scale(f, (max_x, max_y)) for f in features]
    lvl_min = -torch.log2(torch.tensor(scales[0], dtype=torch.float32)).item()
    lvl_max = -torch.log2(torch.tensor(scales[-1], dtype=torch.float32)).item()
    return scales, LevelMapper(int(lvl_min), int(lvl_max), canonical_scale=canonical_scale,
                               canonical_level=canonical_level)


def _multiscale_roi_align(x_filtered: List[Tensor], boxes: List[Tensor], output_size, sampling_ratio: int,
                          scales: Optional[List[float]], mapper: Optional[LevelMapper]) -> Tensor:
    if scales is None or mapper is None:
        raise ValueError("scales and mapper should not be None")
    first = x_filtered[0]
    if (len(x_filtered) > 1 and first.is_cuda and first.dtype in (torch.float32, torch.float16, torch.bfloat16) and first.is_contiguous()
            and 1 <= len(boxes) <= 64 and all(b.is_cuda and b.dtype == torch.float32 and not b.requires_grad for b in boxes)):
        # one op for the whole call: the [K,5] rows of convert_boxes_to_roi_format (ops/_utils.py:18-25) are written by the launch-order
        # pre-pass of the multi-scale launch itself (round 6: one launch and one gap less in front of every call)
        return torch.ops.tvmi.multiscale_roi_align_boxes(
            list(x_filtered), list(boxes), [float(s) for s in scales], int(output_size[0]), int(output_size[1]), int(sampling_ratio),
            False, int(mapper.k_min), int(mapper.k_max), float(mapper.s0), float(mapper.lvl0), float(mapper.eps))[0]
    rois = _convert_to_roi_format(boxes)
    if len(x_filtered) == 1:
        return roi_align(x_filtered[0], rois, output_size=output_size, spatial_scale=scales[0],
                         sampling_ratio=sampling_ratio)
    if first.is_cuda and first.dtype in (torch.float32, torch.float16, torch.bfloat16):
        # one launch for all levels: level assignment happens in the kernel, results land
        # directly in the [K, C, PH, PW] output (no torch.where / index_put per level); the
        # registered autograd formula is one launch too (tvmi::multiscale_roi_align_backward)
        return torch.ops.tvmi.multiscale_roi_align(
            list(x_filtered), rois.float(), [float(s) for s in scales], int(output_size[0]),
            int(output_size[1]), int(sampling_ratio), False, int(mapper.k_min), int(mapper.k_max), float(mapper.s0),
            float(mapper.lvl0), float(mapper.eps))
    levels = mapper(boxes)
    result = torch.zeros((len(rois), first.shape[1]) + tuple(output_size), dtype=first.dtype, device=first.device)
    for level, (feature, scale) in enumerate(zip(x_filtered, scales)):
        idx = torch.where(levels == level)[0]
        pooled = roi_align(feature, rois[idx], output_size=output_size, spatial_scale=scale,
                           sampling_ratio=sampling_ratio)
        result[idx] = pooled.to(result.dtype)
    return result


class MultiScaleRoIAlign(nn.Module):
    """Multi-scale RoIAlign over a dict of FPN feature maps (torchvision.ops.MultiScaleRoIAlign)."""

    def __init__(self, featmap_names: List[str], output_size: Union[int, Tuple[int], List[int]],
                 sampling_ratio: int, *, canonical_scale: int = 224, canonical_level: int = 4):
        super().__init__()
        if isinstance(output_size, int):
            output_size = (output_size, output_size)
        self.featmap_names, self.sampling_ratio = featmap_names, sampling_ratio
        self.output_size = tuple(output_size)
        self.scales: Optional[List[float]] = None
        self.map_levels: Optional[LevelMapper] = None
        self.canonical_scale, self.canonical_level = canonical_scale, canonical_level

    def forward(self, x: Dict[str, Tensor], boxes: List[Tensor], image_shapes: List[Tuple[int, int]]) -> Tensor:
        x_filtered = [v for k, v in x.items() if k in self.featmap_names]
        if self.scales is None or self.map_levels is None:
            self.scales, self.map_levels = _setup_scales(x_filtered, image_shapes, self.canonical_scale,
                                                         self.canonical_level)
        return _multiscale_roi_align(x_filtered, boxes, self.output_size, self.sampling_ratio, self.scales,
                                     self.map_levels)

    def forward_with_nms_step(self, x: Dict[str, Tensor], boxes: List[Tensor], image_shapes: List[Tuple[int, int]], dets: Tensor,
                              scores: Tensor, idxs: Tensor, iou_threshold: float, num_segments: int, image_idx: Tensor,
                              num_images: int, max_dets: int, labels: Optional[Tensor] = None):
        """`forward(x, boxes, image_shapes)` AND `sharding.nms_pack_payload(dets, scores, idxs, ...)` of one detector step as ONE
        call: (`pooled`, `keep`, `num`, `payload`).  The two jobs are independent (RoIAlign reads the maps and `boxes`, the NMS
        `dets` / `scores`); where both take their one-launch kernels (`tvmi::roi_align_boxes_nms_step`: contiguous float32 /
        float16 / bfloat16 CUDA maps of several levels, float32 boxes, <= 4096 float32 `dets` in <= 64 segments and <= 16 images)
        the NMS workgroups ride in front of the RoIAlign grid — one launch on one stream instead of two streams with a fork and
        a join.  Anything else: the two calls one after the other.  Inference only (no autograd through this entry)."""
        from . import sharding
        x_filtered = [v for k, v in x.items() if k in self.featmap_names]
        if self.scales is None or self.map_levels is None:
            self.scales, self.map_levels = _setup_scales(x_filtered, image_shapes, self.canonical_scale, self.canonical_level)
        first, n, m = x_filtered[0], dets.shape[0], self.map_levels
        if (len(x_filtered) > 1 and first.is_cuda and first.dtype in (torch.float32, torch.float16, torch.bfloat16)
                and all(f.is_contiguous() and not f.requires_grad for f in x_filtered) and 1 <= len(boxes) <= 64
                and all(b.is_cuda and b.dtype == torch.float32 and not b.requires_grad for b in boxes)
                and dets.is_cuda and dets.dtype == torch.float32 and scores.dtype == torch.float32
                and 1 <= n <= sharding.STEP_MAX_BOXES and 1 <= num_segments <= sharding.STEP_MAX_SEGMENTS
                and 1 <= num_images <= sharding.STEP_MAX_IMAGES):
            pooled, _, keep, num, payload = torch.ops.tvmi.roi_align_boxes_nms_step(
                list(x_filtered), list(boxes), [float(s) for s in self.scales], int(self.output_size[0]), int(self.output_size[1]),
                int(self.sampling_ratio), False, int(m.k_min), int(m.k_max), float(m.s0), float(m.lvl0), float(m.eps), dets, scores, idxs,
                float(iou_threshold), int(num_segments), image_idx, labels, int(num_images), int(max_dets))
            return pooled, keep, num, payload
        keep, num, payload = sharding.nms_pack_payload(dets, scores, idxs, iou_threshold, num_segments, image_idx, num_images, max_dets, labels)
        return self.forward(x, boxes, image_shapes), keep, num, payload

    def __repr__(self) -> str:
        return (f"{self.__class__.__name__}(featmap_names={self.featmap_names}, "
                f"output_size={self.output_size}, sampling_ratio={self.sampling_ratio})")
