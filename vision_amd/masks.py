"""Mask post-processing — host-side mirror of torchvision/models/detection/roi_heads.py:378-500.

`paste_masks_in_image` keeps the reference's name, arguments and result, but runs ONE HIP
launch (`tvmi::paste_masks`, csrc/resize.hip) for all detections instead of the reference's
Python loop of `F.interpolate` + `zeros` + slice-assign + `stack`.
"""
from typing import Tuple

import torch
import torch.nn.functional as F
from torch import Tensor

from ._loader import load as _load


def expand_boxes(boxes: Tensor, scale: float) -> Tensor:
    """roi_heads.py:378-395 — grow xyxy boxes about their centre by `scale`."""
    w_half = (boxes[:, 2] - boxes[:, 0]) * 0.5
    h_half = (boxes[:, 3] - boxes[:, 1]) * 0.5
    x_c = (boxes[:, 2] + boxes[:, 0]) * 0.5
    y_c = (boxes[:, 3] + boxes[:, 1]) * 0.5
    w_half = w_half * scale
    h_half = h_half * scale
    return torch.stack([x_c - w_half, y_c - h_half, x_c + w_half, y_c + h_half], dim=1)


def expand_masks(mask: Tensor, padding: int) -> Tuple[Tensor, float]:
    """roi_heads.py:404-413 — zero-pad the masks and return the matching box scale."""
    M = mask.shape[-1]
    return F.pad(mask, (padding,) * 4), float(M + 2 * padding) / M


def paste_masks_in_image(masks: Tensor, boxes: Tensor, img_shape: Tuple[int, int], padding: int = 1) -> Tensor:
    """roi_heads.py:486-500.  masks [N,1,M,M], boxes [N,4] (xyxy, image pixels) -> [N,1,im_h,im_w].

    Device tensors only: the product path is the HIP kernel and there is no CPU fallback."""
    _load()
    im_h, im_w = int(img_shape[0]), int(img_shape[1])
    if masks.dim() != 4 or masks.shape[1] != 1:
        raise ValueError(f"masks should have shape [N, 1, M, M], got {tuple(masks.shape)}")
    if not masks.is_cuda:
        raise RuntimeError("vision_amd.paste_masks_in_image needs device tensors (no CPU fallback in the product path)")
    if masks.shape[0] == 0:
        return masks.new_empty((0, 1, im_h, im_w))
    return torch.ops.tvmi.paste_masks(masks, boxes, im_h, im_w, int(padding))
