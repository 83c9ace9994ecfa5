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


def _torch_choice(k: Sequence[int]) -> int:
    """transform.py:171-177: `random.choice` through torch's RNG — the very call the reference makes, so that a seeded run
    picks the same training sizes."""
    index = int(torch.empty(1).uniform_(0.0, float(len(k))).item())
    return int(k[index])


def resize_boxes(boxes: Tensor, original_size: Sequence[int], new_size: Sequence[int]) -> Tensor:
    """transform.py:303-319, op by op (fp32 ratios as device scalars)."""
    ratio_height, ratio_width = [
        torch.tensor(s, dtype=torch.float32, device=boxes.device) / torch.tensor(s_orig, dtype=torch.float32, device=boxes.device)
        for s, s_orig in zip(new_size, original_size)]
    xmin, ymin, xmax, ymax = boxes.unbind(1)
    return torch.stack((xmin * ratio_width, ymin * ratio_height, xmax * ratio_width, ymax * ratio_height), dim=1)


def resize_keypoints(keypoints: Tensor, original_size: Sequence[int], new_size: Sequence[int]) -> Tensor:
    """transform.py:284-300."""
    ratio_h, ratio_w = [
        torch.tensor(s, dtype=torch.float32, device=keypoints.device) / torch.tensor(s_orig, dtype=torch.float32, device=keypoints.device)
        for s, s_orig in zip(new_size, original_size)]
    out = keypoints.clone()
    out[..., 0] *= ratio_w
    out[..., 1] *= ratio_h
    return out


def transform(images: Sequence[Tensor], targets: Optional[Sequence[dict]] = None, *, training: bool = False,
              min_size=800, max_size: int = 1333, image_mean: Sequence[float] = (0.485, 0.456, 0.406),
              image_std: Sequence[float] = (0.229, 0.224, 0.225), size_divisible: int = 32,
              fixed_size: Optional[Tuple[int, int]] = None, skip_resize: bool = False):
    """`GeneralizedRCNNTransform.forward` (transform.py:119-204) with targets, training mode included:
    every image draws its `min_size` from the list through torch's RNG (`torch_choice`, :171-177), the boxes / keypoints of
    its target are scaled by the fp32 ratios of the new to the old size (:196-204), its masks are resized like the reference
    does (`F.interpolate(mask[:, None].float(), ...)` = nearest, then `.byte()`, :76-82) — on the resize kernels of this
    library — and normalize + bilinear resize + zero-padded batching of the images stay ONE launch for the whole batch.
    Returns (tensors [B, C, Hp, Wp], image_sizes, targets) — the `ImageList` fields and the transformed copies of the
    targets (the input dicts are not modified)."""
    from .resize import interpolate

    images = list(images)
    sizes_min = list(min_size) if isinstance(min_size, (list, tuple)) else [int(min_size)]
    if targets is not None and len(targets) != len(images):
        raise ValueError("targets must hold one dict per image")
    out_targets = None if targets is None else [dict(t) for t in targets]
    per_image_min = []
    for _ in images:
        if training:
            per_image_min.append(None if skip_resize else _torch_choice(sizes_min))
        else:
            per_image_min.append(sizes_min[-1])
    new_sizes = []
    for img, ms in zip(images, per_image_min):
        h, w = int(img.shape[-2]), int(img.shape[-1])
        new_sizes.append((h, w) if ms is None else resized_size(h, w, ms, int(max_size), fixed_size))
    # images: one launch (per-image output sizes are arguments of the kernel)
    _load()
    for img in images:
        if img.dim() != 3:
            raise ValueError(f"images is expected to be a list of 3d tensors of shape [C, H, W], got {img.shape}")
        if not img.is_floating_point():
            raise TypeError(f"Expected input images to be of floating type (in range [0, 1]), but found type {img.dtype} instead")
        if not img.is_cuda:
            raise RuntimeError("vision_amd.transform needs device tensors (no CPU fallback in the product path)")
    stride = float(size_divisible)
    hp = int(math.ceil(float(max(s[0] for s in new_sizes)) / stride) * stride)
    wp = int(math.ceil(float(max(s[1] for s in new_sizes)) / stride) * stride)
    chunks = []
    for i0 in range(0, len(images), 64):
        chunk, cs = images[i0:i0 + 64], new_sizes[i0:i0 + 64]
        chunks.append(torch.ops.tvmi.normalize_resize_batch(chunk, [s[0] for s in cs], [s[1] for s in cs],
                                                            [float(m) for m in image_mean], [float(s) for s in image_std], hp, wp))
    tensors = chunks[0] if len(chunks) == 1 else torch.cat(chunks)
    if out_targets is not None:
        for img, t, ns, ms in zip(images, out_targets, new_sizes, per_image_min):
            if ms is None:
                continue
            old = (int(img.shape[-2]), int(img.shape[-1]))
            if "masks" in t and t["masks"].numel() > 0:
                with torch.no_grad():
                    t["masks"] = interpolate(t["masks"][:, None].float(), size=list(ns), mode="nearest")[:, 0].byte()
            elif "masks" in t:
                t["masks"] = t["masks"].new_zeros((0, ns[0], ns[1]), dtype=torch.uint8)
            t["boxes"] = resize_boxes(t["boxes"], old, ns)
            if "keypoints" in t:
                t["keypoints"] = resize_keypoints(t["keypoints"], old, ns)
    return tensors, new_sizes, out_targets
