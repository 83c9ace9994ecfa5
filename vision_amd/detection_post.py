"""Detector post-processing, batched over the images of a step.

Host-side mirrors of
  * `RegionProposalNetwork.filter_proposals`   torchvision/models/detection/rpn.py:231-286
  * `RoIHeads.postprocess_detections`          torchvision/models/detection/roi_heads.py:680-737
with the same argument meaning and the same per-image list results, but run as
  [torch.topk per level]  ->  ONE candidate kernel for the whole batch (tvmi::rpn_candidates /
  tvmi::detection_candidates: gather / softmax-sigmoid / BoxCoder.decode / clip / filters)
  ->  ONE masked segmented NMS over all images (tvmi::nms_segmented_masked, segment = image x level|class:
      filtered-out candidates leave the problem ON THE DEVICE — score -inf / largest key put them behind every live
      candidate in both sorts and the kernels read the live count from memory; no `nonzero`, no gathers)
  ->  ONE top-k packing launch (tvmi::pack_detections_devcount, keep length read on the device).
`padded=True` returns the fixed-shape payload ([B, max, 6] + counts) without the final split, which
is what `vision_amd.sharding.all_gather_detections` ships between GPUs — and that form has NO host
synchronisation at all (it runs under `torch.cuda.set_sync_debug_mode("error")`, tests/test_gpu_parity.py); the
list-returning form reads the per-image counts once, at the very end (the reference takes ~6 reads per image).
Device tensors only — there is no CPU fallback in the product path.
"""
import math
from typing import List, Sequence, Tuple

import torch
from torch import Tensor

from ._loader import load as _load

BBOX_XFORM_CLIP = math.log(1000.0 / 16)  # models/detection/_utils.py:141


def _host_to_device(values, dtype, device) -> Tensor:
    """Small host-built table -> device without a blocking copy (pinned staging + non_blocking: a pageable
    `torch.tensor(..., device=)` is a synchronising memcpy, which the sync-free forms must not contain)."""
    return torch.tensor(values, dtype=dtype).pin_memory().to(device, non_blocking=True)


def _image_hw(image_shapes: Sequence[Tuple[int, int]], device) -> Tensor:
    return _host_to_device([[float(h), float(w)] for h, w in image_shapes], torch.float32, device)


def _need_cuda(t: Tensor, what: str):
    if not t.is_cuda:
        raise RuntimeError(f"vision_amd.{what} needs device tensors (no CPU fallback in the product path)")


# limits of the device-count NMS forms (nms.hip: capacity check of nms_segmented_entry; per-segment limit of the banded
# segment-major sweep).  Up to these the masked op cannot fail; beyond them `_masked_nms` compacts first.
NMS_CAPACITY = 1_200_000
SEGMENT_LIMIT = 8192


def _masked_nms(b: Tensor, s: Tensor, seg: Tensor, valid: Tensor, thresh: float, num_segments: int, max_per_segment: int):
    """(keep, num) of the class/level-segmented NMS over the live candidates.  `max_per_segment`: the caller's static bound on
    the candidates of one segment.  Within the limits of the device-count kernels this is ONE masked op with no host read;
    beyond them (a candidate grid above 1.2 M entries — Faster R-CNN heads with ~1200 classes, 91 classes from batch 14; or a
    segment that may exceed 8,192 boxes — rpn_pre_nms_top_n > 8192) the live candidates are compacted first (`nonzero`: one
    host read) and the general path of tvmi::nms_segmented takes them.  Never a silently wrong list."""
    n = b.shape[0]
    if n <= NMS_CAPACITY and min(n, max_per_segment) <= SEGMENT_LIMIT:
        return torch.ops.tvmi.nms_segmented_masked(b, s, seg, valid, thresh, num_segments, int(max_per_segment))
    live = torch.nonzero(valid).squeeze(1)
    keep = live[torch.ops.tvmi.nms_segmented(b[live], s[live], seg[live], thresh, -1)]
    return keep, torch.full((1,), keep.numel(), dtype=torch.int64, device=b.device)


def _split(dets: Tensor, counts: Tensor, with_labels: bool):
    boxes, scores, labels = [], [], []
    for d, n in zip(dets, counts.tolist()):
        if n < 0:
            raise RuntimeError("vision_amd: the segmented NMS reported a segment beyond its static bound "
                               "(tvmi::nms_segmented_masked returned -1); no detections were produced")
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
    offsets = _host_to_device([sum(num_anchors_per_level[:i]) for i in range(L)], torch.int64, proposals.device)
    boxes, scores, levels, valid = torch.ops.tvmi.rpn_candidates(
        objectness, proposals.reshape(B, A, 4), None if pred_bbox_deltas is None else pred_bbox_deltas.detach().reshape(B, A, 4),
        top_idx, offsets, _image_hw(image_shapes, proposals.device), BBOX_XFORM_CLIP, float(score_thresh), float(min_size))
    img = torch.arange(B, device=proposals.device, dtype=torch.int64).repeat_interleave(T)          # image of candidate i (static shape)
    b, s = boxes.reshape(-1, 4), scores.reshape(-1)
    per_level = max(min(int(pre_nms_top_n), int(n)) for n in num_anchors_per_level)
    keep, num = _masked_nms(b, s, img * L + levels.reshape(-1), valid.reshape(-1), float(nms_thresh), B * L, per_level)
    dets, counts = torch.ops.tvmi.pack_detections_devcount(b, s, None, img, keep, num, B, int(post_nms_top_n))
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
    # image of every RoI row, built ON THE DEVICE with a host-known output size (no synchronisation).  Not on the host:
    # aten's CPU repeat_interleave is an OpenMP parallel_for with grain size 1 — its worker threads spin after the region,
    # and inside a CPU-quota'd container that burns the process's CFS budget: measured on the GPU box as a 45-60 ms stop of
    # the whole process in every third Mask R-CNN step (profiles/r05_e2e_stall.md)
    sizes = [int(p.shape[0]) for p in proposals]
    row_image = torch.repeat_interleave(torch.arange(B, device=dev, dtype=torch.int32), _host_to_device(sizes, torch.int64, dev),
                                        output_size=sum(sizes))
    cb, cs, cv = torch.ops.tvmi.detection_candidates(
        class_logits, box_regression, torch.cat(list(proposals), 0), row_image, _image_hw(image_shapes, dev),
        [float(w) for w in bbox_reg_weights], BBOX_XFORM_CLIP, float(score_thresh), 1e-2)
    R = class_logits.shape[0]
    # candidate (r, c) of the [R, C-1] grid: label c + 1, image of row r — static-shape index arithmetic, no compaction
    labels = torch.arange(1, C, device=dev, dtype=torch.int64).repeat(R)
    img = row_image.to(torch.int64).repeat_interleave(C - 1)
    b, s = cb.reshape(-1, 4), cs.reshape(-1)
    keep, num = _masked_nms(b, s, img * C + labels, cv.reshape(-1), float(nms_thresh), B * C, max(sizes, default=0))
    dets, counts = torch.ops.tvmi.pack_detections_devcount(b, s, labels, img, keep, num, B, int(detections_per_img))
    return (dets, counts) if padded else _split(dets, counts, True)


def retinanet_postprocess_detections(cls_logits: Sequence[Tensor], bbox_regression: Sequence[Tensor], anchors: Sequence[Sequence[Tensor]],
                                     image_shapes: Sequence[Tuple[int, int]], *, score_thresh: float = 0.05,
                                     topk_candidates: int = 1000, nms_thresh: float = 0.5, detections_per_img: int = 300,
                                     padded: bool = False):
    """`RetinaNet.postprocess_detections` (torchvision/models/detection/retinanet.py:509-571), batched over the images of a step.
    `cls_logits[l]` [B, A_l, K] and `bbox_regression[l]` [B, A_l, 4] per pyramid level (what the reference's forward splits the
    head outputs into, :624-636), `anchors[i][l]` [A_l, 4] per image and level.  The reference loops images x levels (sigmoid,
    threshold, top-k, decode, clip: ~12 launches each) and calls batched_nms per image; here: per level ONE sigmoid + top-k over
    the batch (library plumbing), ONE gather / decode / clip kernel for all survivors (`tvmi::rpn_candidates`: BoxCoder weights
    (1, 1, 1, 1) and the clip of `det_utils.BoxCoder`, the same as the RPN's), ONE masked class-segmented NMS over all
    images (below-threshold candidates leave on the device) and one packing launch.  Returns the reference's list of {boxes, scores, labels} dicts (or the padded payload)."""
    _load()
    _need_cuda(cls_logits[0], "retinanet_postprocess_detections")
    B, K = cls_logits[0].shape[0], cls_logits[0].shape[-1]
    dev = cls_logits[0].device
    cand_logit, cand_anchor, cand_delta, cand_label, cand_ok, per_level = [], [], [], [], [], []
    for lvl, (logits, deltas) in enumerate(zip(cls_logits, bbox_regression)):
        flat = logits.detach().reshape(B, -1)                         # [B, A_l * K], (anchor, class) fastest = class
        scores = torch.sigmoid(flat)
        k = min(int(topk_candidates), flat.shape[1])                  # det_utils._topk_min
        masked = torch.where(scores > score_thresh, scores, scores.new_full((), -1.0))
        vals, idx = masked.topk(k, dim=1)
        a_idx = torch.div(idx, K, rounding_mode="floor")
        anc = torch.stack([anchors[i][lvl] for i in range(B)])        # [B, A_l, 4]
        gi = a_idx[..., None].expand(-1, -1, 4)
        cand_logit.append(flat.gather(1, idx))
        cand_anchor.append(anc.gather(1, gi))
        cand_delta.append(deltas.detach().gather(1, gi))
        cand_label.append(idx - a_idx * K)
        cand_ok.append(vals > score_thresh)
        per_level.append(k)
    logit, anc, dlt = torch.cat(cand_logit, 1), torch.cat(cand_anchor, 1).contiguous(), torch.cat(cand_delta, 1).contiguous()
    labels, ok = torch.cat(cand_label, 1), torch.cat(cand_ok, 1)
    T = logit.shape[1]
    top_idx = torch.arange(T, device=dev, dtype=torch.int64).expand(B, T).contiguous()
    offsets = _host_to_device([sum(per_level[:i]) for i in range(len(per_level))], torch.int64, dev)
    boxes, scores, _, _ = torch.ops.tvmi.rpn_candidates(logit.contiguous(), anc, dlt, top_idx, offsets, _image_hw(image_shapes, dev),
                                                        BBOX_XFORM_CLIP, 0.0, -1.0)
    img = torch.arange(B, device=dev, dtype=torch.int64).repeat_interleave(T)
    b, sc, lab = boxes.reshape(-1, 4), scores.reshape(-1), labels.reshape(-1)
    keep, num = _masked_nms(b, sc, img * K + lab, ok.reshape(-1), float(nms_thresh), B * K, T)
    dets, counts = torch.ops.tvmi.pack_detections_devcount(b, sc, lab, img, keep, num, B, int(detections_per_img))
    if padded:
        return dets, counts
    bl, sl, ll = _split(dets, counts, True)
    return [{"boxes": x, "scores": y, "labels": z} for x, y, z in zip(bl, sl, ll)]
