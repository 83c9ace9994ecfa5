"""Input transform of the detection models — host-side mirror of
`GeneralizedRCNNTransform.forward` (torchvision/models/detection/transform.py:119-255), inference form.

Same size rules as the reference (scale = min(min_size / min(h, w), max_size / max(h, w)),
`F.interpolate(..., recompute_scale_factor=True)` output size = floor(size * scale), zero-padded batch
with both sides rounded up to `size_divisible`), but normalize + bilinear resize + batching run as ONE
HIP launch for the whole batch (`tvmi::normalize_resize_batch`) instead of ~6 launches and two
host-to-device copies per image.  Device tensors only (no CPU fallback in the product path).
"""
import math
from typing import List, Optional, Sequence, Tuple

import torch
from torch import Tensor

from ._loader import load as _load


def resized_size(h: int, w: int, min_size: int, max_size: int, fixed_size: Optional[Tuple[int, int]] = None) -> Tuple[int, int]:
    """transform.py:25-72: the (height, width) `_resize_image_and_masks` produces for an h x w image."""
    if fixed_size is not None:
        return int(fixed_size[1]), int(fixed_size[0])
    scale = min(float(min_size) / float(min(h, w)), float(max_size) / float(max(h, w)))
    return int(math.floor(float(h) * scale)), int(math.floor(float(w) * scale))


def transform_images(images: Sequence[Tensor], min_size: int = 800, max_size: int = 1333,
                     image_mean: Sequence[float] = (0.485, 0.456, 0.406), image_std: Sequence[float] = (0.229, 0.224, 0.225),
                     size_divisible: int = 32, fixed_size: Optional[Tuple[int, int]] = None) -> Tuple[Tensor, List[Tuple[int, int]]]:
    """-> (`ImageList.tensors` [B, C, Hp, Wp], `ImageList.image_sizes`) exactly like the reference in eval mode
    (for several `min_size` values the reference's eval branch uses the last one, transform.py:186)."""
    _load()
    images = list(images)
    if not images:
        raise ValueError("images list should not be empty")
    for img in images:
        if img.dim() != 3:
            raise ValueError(f"images is expected to be a list of 3d tensors of shape [C, H, W], got {img.shape}")
        if not img.is_floating_point():
            raise TypeError(f"Expected input images to be of floating type (in range [0, 1]), but found type {img.dtype} instead")
        if not img.is_cuda:
            raise RuntimeError("vision_amd.transform_images needs device tensors (no CPU fallback in the product path)")
    if isinstance(min_size, (list, tuple)):
        min_size = min_size[-1]
    sizes = [resized_size(int(i.shape[-2]), int(i.shape[-1]), int(min_size), int(max_size), fixed_size) for i in images]
    stride = float(size_divisible)
    hp = int(math.ceil(float(max(s[0] for s in sizes)) / stride) * stride)
    wp = int(math.ceil(float(max(s[1] for s in sizes)) / stride) * stride)
    out = []
    for i0 in range(0, len(images), 64):   # the launch takes 64 images; bigger batches go in slices
        chunk, cs = images[i0:i0 + 64], sizes[i0:i0 + 64]
        out.append(torch.ops.tvmi.normalize_resize_batch(chunk, [s[0] for s in cs], [s[1] for s in cs],
                                                         [float(m) for m in image_mean], [float(s) for s in image_std], hp, wp))
    return (out[0] if len(out) == 1 else torch.cat(out)), sizes
