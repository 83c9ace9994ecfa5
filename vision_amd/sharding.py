"""Image-level sharding of the detection hot path across the GPUs of one node.

The path shards naturally over images (no cross-image state anywhere between the RPN and the
box head: models/detection/rpn.py:276, roi_heads.py:702 loop per image).  One process per
GPU takes a contiguous slice of the batch, runs RoIAlign + NMS on it, and the only exchange
step is ONE fixed-shape all-gather of the padded detections over RCCL/xGMI
(`all_gather_into_tensor`, 2.4 KB per image) — replacing the pickled `all_gather_object` of
references/detection/utils.py:70-83 (2 collectives + host pickling per call).
Works with the `nccl` (= RCCL on ROCm) and `gloo` backends alike.
"""
from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist
from torch import Tensor

DET_FIELDS = 6  # x1, y1, x2, y2, score, label


def shard_range(num_items: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous, balanced [start, end) slice of `num_items` for `rank` (first ranks get the
    remainder)."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError(f"invalid rank {rank} / world size {world_size}")
    base, rem = divmod(num_items, world_size)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def pack_detections(boxes: Sequence[Tensor], scores: Sequence[Tensor], labels: Sequence[Tensor],
                    max_dets: int) -> Tuple[Tensor, Tensor]:
    """Per-image variable-length detections -> (`dets` [B, max_dets, 6] fp32 zero padded,
    `counts` [B] int32).  Detections beyond max_dets are dropped (they are already sorted by
    score on this path)."""
    B = len(boxes)
    device = boxes[0].device if B else torch.device("cpu")
    dets = torch.zeros((B, max_dets, DET_FIELDS), dtype=torch.float32, device=device)
    counts = torch.zeros((B,), dtype=torch.int32, device=device)
    for i, (b, s, l) in enumerate(zip(boxes, scores, labels)):
        n = min(int(b.shape[0]), max_dets)
        if n:
            dets[i, :n, :4] = b[:n].to(torch.float32)
            dets[i, :n, 4] = s[:n].to(torch.float32)
            dets[i, :n, 5] = l[:n].to(torch.float32)
        counts[i] = n
    return dets, counts


def pack_detection_dicts(outputs: Sequence[dict], max_dets: int) -> Tuple[Tensor, Tensor]:
    """The per-image result dicts of a detection model (`boxes` [n,4], `scores` [n], `labels` [n], already sorted by
    score) -> the fixed-shape all-gather payload.  Uses tensor SHAPES only (host-known): no device value is read, so
    nothing synchronises between the model and the collective."""
    import torch.nn.functional as F

    B = len(outputs)
    device = outputs[0]["boxes"].device if B else torch.device("cpu")
    rows, ns = [], []
    for o in outputs:
        n = min(int(o["boxes"].shape[0]), max_dets)
        r = torch.cat([o["boxes"][:n].to(torch.float32), o["scores"][:n, None].to(torch.float32),
                       o["labels"][:n, None].to(torch.float32)], 1)
        rows.append(F.pad(r, (0, 0, 0, max_dets - n)))
        ns.append(n)
    dets = torch.stack(rows) if B else torch.zeros((0, max_dets, DET_FIELDS), device=device)
    return dets, torch.tensor(ns, dtype=torch.int32).to(device, non_blocking=True)


def pack_kept_detections(boxes: Tensor, scores: Tensor, image_idx: Tensor, keep: Tensor, num_images: int,
                          max_dets: int, labels: Tensor = None, num_keep: Tensor = None) -> Tuple[Tensor, Tensor]:
    """Batched form used on the hot path: `keep` is the score-ordered output of a batched NMS over
    all images of this rank; returns the same (`dets`, `counts`) payload as pack_detections.  On CUDA
    tensors this is ONE launch (`tvmi::pack_detections`).  With `num_keep` (a [1] int64 device tensor,
    from `tvmi::nms_segmented_padded`) only the first num_keep[0] entries of `keep` are read and no host
    synchronisation happens between the NMS and the packing — the chain is hipGraph-capturable.  If the sync-free
    NMS reported its error sentinel (num_keep < 0: a segment above its size limit or an id outside the promised range)
    every count comes back as -1 — `unpack_detections` raises on it — instead of "no detections"."""
    if num_keep is not None:
        return torch.ops.tvmi.pack_detections_devcount(boxes, scores, labels, image_idx, keep, num_keep,
                                                       int(num_images), int(max_dets))
    if boxes.is_cuda and num_images <= 65535:
        return torch.ops.tvmi.pack_detections(boxes, scores, labels, image_idx, keep, int(num_images), int(max_dets))
    ki = image_idx[keep]
    per_b, per_s, per_l = [], [], []
    for i in range(num_images):
        sel = keep[ki == i]
        per_b.append(boxes[sel])
        per_s.append(scores[sel])
        per_l.append(labels[sel] if labels is not None else torch.zeros_like(sel))
    return pack_detections(per_b, per_s, per_l, max_dets)


def pack_kept_payload(boxes: Tensor, scores: Tensor, image_idx: Tensor, keep: Tensor, num_keep: Tensor, num_images: int,
                      max_dets: int, labels: Tensor = None) -> Tensor:
    """The hot-path form of pack_kept_detections for N > 1: ONE launch writes the collective payload itself —
    `[num_images, max_dets * 6 + 1]` fp32, row b = the padded detections of image b followed by its count — so nothing is
    assembled between the NMS and `all_gather_payload` (~6 tiny torch launches per step before)."""
    return torch.ops.tvmi.pack_detections_payload(boxes, scores, labels, image_idx, keep, num_keep, int(num_images), int(max_dets))


# limits of the one-launch step kernel (include/tvmi.h: tvmi_nms_step)
STEP_MAX_BOXES, STEP_MAX_SEGMENTS, STEP_MAX_IMAGES = 4096, 64, 16


def nms_pack_payload(boxes: Tensor, scores: Tensor, idxs: Tensor, iou_threshold: float, num_segments: int, image_idx: Tensor,
                     num_images: int, max_dets: int, labels: Tensor = None) -> Tuple[Tensor, Tensor, Tensor]:
    """`boxes.batched_nms_padded` + `pack_kept_payload` of a detector step: (`keep` [N], `num` [1] on the device, `payload`
    [num_images, max_dets * 6 + 1]).  Up to 4096 float32 boxes in <= 64 segments and <= 16 images this is ONE launch
    (`tvmi::nms_step`: score order, per-segment suppression tiles, sweeps, the global-order keep list and the padded top-k rows are
    phases of one grid — the reference does this with `batched_nms` + a python loop over the images, ops/boxes.py:57-126,
    roi_heads.py:716-737); anything larger takes the two ops.  No host synchronisation either way."""
    n = boxes.shape[0]
    if (boxes.is_cuda and boxes.dtype == torch.float32 and scores.dtype == torch.float32 and 1 <= n <= STEP_MAX_BOXES
            and 1 <= num_segments <= STEP_MAX_SEGMENTS and 1 <= num_images <= STEP_MAX_IMAGES):
        return torch.ops.tvmi.nms_step(boxes, scores, idxs, float(iou_threshold), int(num_segments), image_idx, labels,
                                       int(num_images), int(max_dets))
    keep, num = torch.ops.tvmi.nms_segmented_padded(boxes, scores, idxs, float(iou_threshold), int(num_segments))
    return keep, num, pack_kept_payload(boxes, scores, image_idx, keep, num, num_images, max_dets, labels)


def split_payload(payload: Tensor, max_dets: int) -> Tuple[Tensor, Tensor]:
    """(`dets` [B, max_dets, 6], `counts` [B] float32) VIEWS of a payload — no launch, no copy."""
    return payload[:, : max_dets * DET_FIELDS].unflatten(1, (max_dets, DET_FIELDS)), payload[:, max_dets * DET_FIELDS]


def all_gather_payload(payload: Tensor, max_dets: int, group=None, always_collective: bool = False) -> Tuple[Tensor, Tensor]:
    """The one collective of the path on a payload written by `pack_kept_payload`: one `all_gather_into_tensor`, then views
    of the gathered buffer (`dets` [world*B, D, 6], `counts` [world*B] as float32 — `unpack_detections` rounds them on the
    host).  A world of one returns views of its own payload."""
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or always_collective):
        world = dist.get_world_size(group)
        gathered = torch.empty((world * payload.shape[0], payload.shape[1]), dtype=payload.dtype, device=payload.device)
        dist.all_gather_into_tensor(gathered, payload, group=group)
        payload = gathered
    return split_payload(payload, max_dets)


def all_gather_detections(dets: Tensor, counts: Tensor, group=None, always_collective: bool = False) -> Tuple[Tensor, Tensor]:
    """All-gather equally shaped per-rank (`dets` [B_local, D, 6], `counts` [B_local]) into
    ([world*B_local, D, 6], [world*B_local]) in rank order.  Counts ride in the same buffer as
    the detections (one collective, not two).  A world of one returns its inputs untouched unless
    `always_collective` asks for the real collective (a 1-rank RCCL group still goes through
    `all_gather_into_tensor` on device buffers — what tests/test_gpu_dist.py exercises on a 1-GPU box)."""
    if not (dist.is_available() and dist.is_initialized()):
        return dets, counts
    if dist.get_world_size(group) == 1 and not always_collective:
        return dets, counts
    world = dist.get_world_size(group)
    B, D, Fd = dets.shape
    payload = torch.empty((B, D * Fd + 1), dtype=torch.float32, device=dets.device)
    payload[:, : D * Fd] = dets.reshape(B, D * Fd)
    payload[:, D * Fd] = counts.to(torch.float32)
    gathered = torch.empty((world * B, D * Fd + 1), dtype=torch.float32, device=dets.device)
    dist.all_gather_into_tensor(gathered, payload.contiguous(), group=group)
    return gathered[:, : D * Fd].reshape(world * B, D, Fd), gathered[:, D * Fd].round().to(torch.int32)


def unpack_detections(dets: Tensor, counts: Tensor) -> List[dict]:
    out = []
    for d, n in zip(dets, counts.tolist()):
        n = int(round(n))      # counts gathered inside the float payload arrive as float32
        if n < 0:
            raise RuntimeError("detection payload carries the sync-free NMS error sentinel (count -1): a segment exceeded the "
                               "size limit of the no-sync path or an id was outside the promised range; use batched_nms()")
        d = d[:n]
        out.append({"boxes": d[:, :4], "scores": d[:, 4], "labels": d[:, 5].to(torch.int64)})
    return out
