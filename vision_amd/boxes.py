"""Box operators — host-side mirror of torchvision/ops/boxes.py (nms :20-54, batched_nms
:57-126, box helpers :129-391) on top of the gfx950 kernels.  Same names, argument meaning,
return conventions and error behaviour as the reference; no CPU/eager fallback for CUDA
tensors (CPU tensors go wherever the dispatcher has a CPU kernel registered — in this repo
that is only the reference oracle, loaded by tests)."""
from typing import Tuple

import torch
from torch import Tensor

from ._loader import assert_has_ops


def nms(boxes: Tensor, scores: Tensor, iou_threshold: float) -> Tensor:
    """Greedy NMS; int64 indices of kept boxes in decreasing score order
    (torchvision.ops.nms, ops/boxes.py:20-54)."""
    assert_has_ops()
    if boxes.is_quantized:      # ops/boxes.py:48-53: quantized boxes / scores -> qnms on their integer representation
        return torch.ops.torchvision.qnms(boxes.int_repr(), scores.int_repr(), iou_threshold)
    return torch.ops.torchvision.nms(boxes, scores, iou_threshold)


def batched_nms(boxes: Tensor, scores: Tensor, idxs: Tensor, iou_threshold: float, num_segments: int = -1) -> Tensor:
    """NMS that never suppresses across categories (ops/boxes.py:57-91).

    The reference switches between shifting the boxes per category and running one nms()
    ("coordinate trick", :93-109; device tensors up to 100,000 elements) and a python loop over
    categories (:113-126).  On device tensors both run as ONE segment-major launch chain
    (`tvmi::nms_segmented`): category-mismatched pairs are never even tested, every category is swept
    by its own workgroup, and there is no torch.unique / torch.where host round trip per category.
    The ARITHMETIC follows the reference's switch: in the coordinate-trick regime the IoUs are
    evaluated on the shifted boxes `boxes + idxs * (boxes.max() + 1)` (fp32 rounding of the shifted
    coordinates moves IoUs at the threshold edge, so the kept set can differ from the per-category
    loop's — tests/test_gpu_parity.py pins a case), above it on the unshifted boxes (the loop's
    arithmetic).  The trick's one side effect is reproduced too: with a coordinate below -1 the shifted
    categories overlap (max - min > max + 1) and the reference suppresses ACROSS categories — the same
    launch that takes the maximum takes the minimum (`aminmax`), and such an input is answered by the
    reference's own formulation, one global-order `nms()` of the shifted boxes (a second pass: the flag is
    read after the segmented result's size synchronisation, so the usual input pays one small
    device-to-host read, no extra launch).  16-bit boxes shift in their own dtype like the reference
    (`idxs.to(boxes)`; a product beyond the fp16 range is inf there as well).
    `num_segments` (extension, optional): a promise that 0 <= idxs < num_segments, which lets
    N <= 4096 run as a single launch.  CPU tensors (tests only) follow the reference's switch."""
    if boxes.is_cuda:
        assert_has_ops()
        if boxes.numel() == 0:
            return torch.empty((0,), dtype=torch.int64, device=boxes.device)
        if boxes.numel() <= 100_000:   # ops/boxes.py:83: the reference's coordinate-trick regime on device tensors
            min_coordinate, max_coordinate = torch.aminmax(boxes)
            offsets = idxs.to(boxes) * (max_coordinate + torch.tensor(1).to(boxes))
            shifted = boxes + offsets[:, None]
            if torch.compiler.is_compiling() or torch.jit.is_tracing():
                # a traced / compiled graph must stay branch-free like the reference's: its own formulation,
                # one global-order nms() of the shifted boxes — the same result, cross-category side effect included
                return torch.ops.torchvision.nms(shifted, scores, iou_threshold)
            # ONE host read for both data-dependent facts (round 5 read them separately): the length of the
            # keep list (the reference reads it too: nms returns a tensor of that size) and whether the shifted categories overlap
            keep, num = torch.ops.tvmi.nms_segmented_padded(shifted, scores, idxs, iou_threshold, int(num_segments))
            n, overlap = torch.stack([num[0], (min_coordinate < -1).to(torch.int64)]).tolist()
            if overlap:                        # shifted categories overlap: the reference's nms() sees cross-category pairs
                return torch.ops.torchvision.nms(shifted, scores, iou_threshold)
            if n < 0:                          # beyond the limits of the device-count paths (a segment above 8192 boxes ...)
                return torch.ops.tvmi.nms_segmented(shifted, scores, idxs, iou_threshold, int(num_segments))
            return keep[:n]
        return torch.ops.tvmi.nms_segmented(boxes, scores, idxs, iou_threshold, int(num_segments))
    if boxes.numel() > 4000:
        return _batched_nms_vanilla(boxes, scores, idxs, iou_threshold)
    return _batched_nms_coordinate_trick(boxes, scores, idxs, iou_threshold)


def batched_nms_padded(boxes: Tensor, scores: Tensor, idxs: Tensor, iou_threshold: float,
                       num_segments: int = -1) -> Tuple[Tensor, Tensor]:
    """batched_nms without the host synchronisation on the result size (device tensors only): returns
    (`keep` [N] int64 whose first `num[0]` entries are the reference's result, `num` [1] int64 on the device;
    -1 if the input broke the promised limits).  Feed both to
    `vision_amd.sharding.pack_kept_detections(..., num_keep=num)`.
    `num_segments` > 0 promises 0 <= idxs < num_segments; for N <= 4096 the whole batched NMS is then one launch."""
    assert_has_ops()
    return torch.ops.tvmi.nms_segmented_padded(boxes, scores, idxs, iou_threshold, int(num_segments))


def _batched_nms_coordinate_trick(boxes: Tensor, scores: Tensor, idxs: Tensor, iou_threshold: float) -> Tensor:
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    max_coordinate = boxes.max()
    offsets = idxs.to(boxes) * (max_coordinate + torch.tensor(1).to(boxes))
    return nms(boxes + offsets[:, None], scores, iou_threshold)


def _batched_nms_vanilla(boxes: Tensor, scores: Tensor, idxs: Tensor, iou_threshold: float) -> Tensor:
    if boxes.is_cuda:
        assert_has_ops()
        return torch.ops.tvmi.nms_segmented(boxes, scores, idxs, iou_threshold)
    keep_mask = torch.zeros_like(scores, dtype=torch.bool)
    for class_id in torch.unique(idxs):
        curr = torch.where(idxs == class_id)[0]
        keep_mask[curr[nms(boxes[curr], scores[curr], iou_threshold)]] = True
    keep = torch.where(keep_mask)[0]
    return keep[scores[keep].sort(descending=True)[1]]


def _upcast(t: Tensor) -> Tensor:
    # ops/_utils.py:72-84: protect products from overflow
    if t.is_floating_point():
        return t if t.dtype in (torch.float32, torch.float64) else t.float()
    return t if t.dtype in (torch.int32, torch.int64) else t.int()


def box_area(boxes: Tensor) -> Tensor:
    boxes = _upcast(boxes)
    return (boxes[..., 2] - boxes[..., 0]) * (boxes[..., 3] - boxes[..., 1])


def _box_inter_union(boxes1: Tensor, boxes2: Tensor) -> Tuple[Tensor, Tensor]:
    area1, area2 = box_area(boxes1), box_area(boxes2)
    lt = torch.max(boxes1[..., :, None, :2], boxes2[..., None, :, :2])
    rb = torch.min(boxes1[..., :, None, 2:], boxes2[..., None, :, 2:])
    wh = _upcast(rb - lt).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    return inter, area1[..., :, None] + area2[..., None, :] - inter


def box_iou(boxes1: Tensor, boxes2: Tensor, fmt: str = "xyxy") -> Tensor:
    """Pairwise IoU [N, M].  Axis-aligned formats are plain tensor math exactly as in the
    reference (ops/boxes.py:314-391); rotated `cxcywhr` boxes go to the
    `torchvision::box_iou_rotated` kernel (:393-399)."""
    if fmt == "cxcywhr":
        assert_has_ops()
        return torch.ops.torchvision.box_iou_rotated(boxes1, boxes2)
    if fmt != "xyxy":
        raise ValueError(f"Unsupported box format {fmt!r}; convert to xyxy or cxcywhr first")
    if _native_pairwise(boxes1, boxes2):
        return torch.ops.tvmi.box_iou_pairwise(boxes1, boxes2, 0)
    inter, union = _box_inter_union(boxes1, boxes2)
    return inter / union


def _native_pairwise(boxes1: Tensor, boxes2: Tensor) -> bool:
    # one launch on device tensors (the matcher's use, roi_heads.py:595 / rpn.py:167, runs without grad)
    return (boxes1.is_cuda and boxes2.is_cuda and boxes1.dim() == 2 and boxes2.dim() == 2 and boxes1.is_floating_point()
            and boxes2.is_floating_point() and not (torch.is_grad_enabled() and (boxes1.requires_grad or boxes2.requires_grad)))


def generalized_box_iou(boxes1: Tensor, boxes2: Tensor) -> Tensor:
    """Pairwise generalized IoU [N, M] of xyxy boxes (ops/boxes.py:409-436)."""
    if _native_pairwise(boxes1, boxes2):
        return torch.ops.tvmi.box_iou_pairwise(boxes1, boxes2, 1)
    inter, union = _box_inter_union(boxes1, boxes2)
    iou = inter / union
    lti = torch.min(boxes1[..., :, None, :2], boxes2[..., None, :, :2])
    rbi = torch.max(boxes1[..., :, None, 2:], boxes2[..., None, :, 2:])
    whi = _upcast(rbi - lti).clamp(min=0)
    areai = whi[..., 0] * whi[..., 1]
    return iou - (areai - union) / areai


def _box_diou_iou(boxes1: Tensor, boxes2: Tensor, eps: float = 1e-7) -> Tuple[Tensor, Tensor]:
    # ops/boxes.py:494-515 (tensor-math form: CPU tensors and inputs that need grad)
    inter, union = _box_inter_union(boxes1, boxes2)
    iou = inter / union
    lti = torch.min(boxes1[..., :, None, :2], boxes2[..., None, :, :2])
    rbi = torch.max(boxes1[..., :, None, 2:], boxes2[..., None, :, 2:])
    whi = _upcast(rbi - lti).clamp(min=0)
    diagonal_distance_squared = (whi[..., 0] ** 2) + (whi[..., 1] ** 2) + eps
    x_p, y_p = (boxes1[..., 0] + boxes1[..., 2]) / 2, (boxes1[..., 1] + boxes1[..., 3]) / 2
    x_g, y_g = (boxes2[..., 0] + boxes2[..., 2]) / 2, (boxes2[..., 1] + boxes2[..., 3]) / 2
    centers_distance_squared = (_upcast(x_p[..., :, None] - x_g[..., None, :]) ** 2) + (
        _upcast(y_p[..., :, None] - y_g[..., None, :]) ** 2)
    return iou - (centers_distance_squared / diagonal_distance_squared), iou


def distance_box_iou(boxes1: Tensor, boxes2: Tensor, eps: float = 1e-7) -> Tensor:
    """Pairwise distance IoU [N, M] of xyxy boxes (ops/boxes.py:469-491); one launch on device tensors."""
    boxes1, boxes2 = _upcast(boxes1), _upcast(boxes2)
    if _native_pairwise(boxes1, boxes2):
        return torch.ops.tvmi.box_iou_pairwise(boxes1, boxes2, 2, float(eps))
    return _box_diou_iou(boxes1, boxes2, eps)[0]


def complete_box_iou(boxes1: Tensor, boxes2: Tensor, eps: float = 1e-7) -> Tensor:
    """Pairwise complete IoU [N, M] of xyxy boxes (ops/boxes.py:439-466); one launch on device tensors."""
    boxes1, boxes2 = _upcast(boxes1), _upcast(boxes2)
    if _native_pairwise(boxes1, boxes2):
        return torch.ops.tvmi.box_iou_pairwise(boxes1, boxes2, 3, float(eps))
    diou, iou = _box_diou_iou(boxes1, boxes2, eps)
    w_pred, h_pred = boxes1[..., :, None, 2] - boxes1[..., :, None, 0], boxes1[..., :, None, 3] - boxes1[..., :, None, 1]
    w_gt, h_gt = boxes2[..., None, :, 2] - boxes2[..., None, :, 0], boxes2[..., None, :, 3] - boxes2[..., None, :, 1]
    v = (4 / (torch.pi ** 2)) * torch.pow(torch.atan(w_pred / h_pred) - torch.atan(w_gt / h_gt), 2)
    with torch.no_grad():
        alpha = v / (1 - iou + v + eps)
    return diou - alpha * v


def clip_boxes_to_image(boxes: Tensor, size: Tuple[int, int]) -> Tensor:
    height, width = size
    x = boxes[..., 0::2].clamp(min=0, max=width)
    y = boxes[..., 1::2].clamp(min=0, max=height)
    return torch.stack((x, y), dim=boxes.dim()).reshape(boxes.shape)


def remove_small_boxes(boxes: Tensor, min_size: float) -> Tensor:
    ws, hs = boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]
    return torch.where((ws >= min_size) & (hs >= min_size))[0]
