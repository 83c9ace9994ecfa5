"""Detector post-processing, batched over the images of a step.

Host-side mirrors of
  * `RegionProposalNetwork.filter_proposals`   torchvision/models/detection/rpn.py:231-286
  * `RoIHeads.postprocess_detections`          torchvision/models/detection/roi_heads.py:680-737
with the same argument meaning and the same per-image list results, but run as
  [torch.topk per level]  ->  ONE candidate kernel for the whole batch (tvmi::rpn_candidates /
  tvmi::detection_candidates: gather / softmax-sigmoid / BoxCoder.decode / clip / filters)
  ->  one `nonzero` (the only host sync; the reference takes ~6 per image)
  ->  ONE segmented NMS over all images (tvmi::nms_segmented, segment = image x level|class)
  ->  ONE top-k packing launch (tvmi::pack_detections).
`padded=True` returns the fixed-shape payload ([B, max, 6] + counts) without the final split, which
is what `vision_amd.sharding.all_gather_detections` ships between GPUs.
Device tensors only — there is no CPU fallback in the product path.
"""
import math
from typing import List, Sequence, Tuple

import torch
from torch import Tensor

from ._loader import load as _load

BBOX_XFORM_CLIP = math.log(1000.0 / 16)  # models/detection/_utils.py:141


def _image_hw(image_shapes: Sequence[Tuple[int, int]], device) -> Tensor:
    return torch.tensor([[float(h), float(w)] for h, w in image_shapes], dtype=torch.float32, device=device)


def _need_cuda(t: Tensor, what: str):
    if not t.is_cuda:
        raise RuntimeError(f"vision_amd.{what} needs device tensors (no CPU fallback in the product path)")


def _split(dets: Tensor, counts: Tensor, with_labels: bool):
    boxes, scores, labels = [], [], []
    for d, n in zip(dets, counts.tolist()):
        d = d[:n]
        boxes.append(d[:, :4])
        scores.append(d[:, 4])
        labels.append(d[:, 5].to(torch.int64))
    return (boxes, scores, labels) if with_labels else (boxes, scores)


def filter_proposals(proposals: Tensor, objectness: Tensor, image_shapes: Sequence[Tuple[int, int]],
                     num_anchors_per_level: Sequence[int], *, pre_nms_top_n: int, post_nms_top_n: int,
                     nms_thresh: float = 0.7, score_thresh: float = 0.0, min_size: float = 1e-3,
                     pred_bbox_deltas: Tensor = None, padded: bool = False):
    """rpn.py:242-286.  `proposals` [B, A, 4] decoded boxes — or, with `pred_bbox_deltas` [B, A, 4] given, the
    anchors, in which case only the per-level top-k survivors are decoded (rpn.py:364-366 decodes all A).
    Returns (list of boxes [n_i, 4], list of scores [n_i]) like the reference."""
    _load()
    _need_cuda(proposals, "filter_proposals")
    B = proposals.shape[0]
    objectness = objectness.detach().reshape(B, -1)
    A = objectness.shape[1]
    # per-level top-k (rpn.py:231-240); torch.topk is library plumbing
    idx, off = [], 0
    for n in num_anchors_per_level:
        k = min(int(pre_nms_top_n), int(n))
        idx.append(objectness[:, off:off + n].topk(k, dim=1)[1] + off)
        off += n
    top_idx = torch.cat(idx, dim=1)
    T, L = top_idx.shape[1], len(num_anchors_per_level)
    offsets = torch.tensor([0] + list(num_anchors_per_level[:-1]), dtype=torch.int64).cumsum(0).to(proposals.device)
    boxes, scores, levels, valid = torch.ops.tvmi.rpn_candidates(
        objectness, proposals.reshape(B, A, 4), None if pred_bbox_deltas is None else pred_bbox_deltas.detach().reshape(B, A, 4),
        top_idx, offsets, _image_hw(image_shapes, proposals.device), BBOX_XFORM_CLIP, float(score_thresh), float(min_size))
    sel = valid.reshape(-1).nonzero()[:, 0]
    img = sel // T
    b, s = boxes.reshape(-1, 4)[sel], scores.reshape(-1)[sel]
    keep = torch.ops.tvmi.nms_segmented(b, s, img * L + levels.reshape(-1)[sel], float(nms_thresh), B * L)
    dets, counts = torch.ops.tvmi.pack_detections(b, s, None, img, keep, B, int(post_nms_top_n))
    return (dets, counts) if padded else _split(dets, counts, False)


def postprocess_detections(class_logits: Tensor, box_regression: Tensor, proposals: List[Tensor],
                           image_shapes: Sequence[Tuple[int, int]], *,
                           bbox_reg_weights: Sequence[float] = (10.0, 10.0, 5.0, 5.0), score_thresh: float = 0.05,
                           nms_thresh: float = 0.5, detections_per_img: int = 100, padded: bool = False):
    """roi_heads.py:680-737.  class_logits [R, C], box_regression [R, 4C], proposals: per-image [R_i, 4].
    Returns (boxes, scores, labels) lists like the reference."""
    _load()
    _need_cuda(class_logits, "postprocess_detections")
    B, C = len(proposals), class_logits.shape[-1]
    dev = class_logits.device
    row_image = torch.repeat_interleave(torch.arange(B, device=dev, dtype=torch.int32),
                                        torch.tensor([p.shape[0] for p in proposals], device=dev))
    cb, cs, cv = torch.ops.tvmi.detection_candidates(
        class_logits, box_regression, torch.cat(list(proposals), 0), row_image, _image_hw(image_shapes, dev),
        [float(w) for w in bbox_reg_weights], BBOX_XFORM_CLIP, float(score_thresh), 1e-2)
    sel = cv.reshape(-1).nonzero()[:, 0]
    r = sel // (C - 1)
    labels = sel - r * (C - 1) + 1
    img = row_image[r].to(torch.int64)
    b, s = cb.reshape(-1, 4)[sel], cs.reshape(-1)[sel]
    keep = torch.ops.tvmi.nms_segmented(b, s, img * C + labels, float(nms_thresh), B * C)
    dets, counts = torch.ops.tvmi.pack_detections(b, s, labels, img, keep, B, int(detections_per_img))
    return (dets, counts) if padded else _split(dets, counts, True)
