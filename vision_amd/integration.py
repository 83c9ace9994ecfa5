"""Laying the reference's python package over this operator library.

torchvision/extension.py:8-33 looks for `_C` and `_C_stable` shared objects NEXT TO the
package and `torch.ops.load_library`s them.  `make_overlay` builds a directory holding a
`torchvision/` whose entries are symlinks to an existing reference checkout / install plus
`_C.so` -> our `tvmi_torch.so` and `_C_stable.so` -> our `tvmi_torch_stable.so`; put it first on sys.path and
`import torchvision` — nothing of the reference is copied or modified.  Set
TVMI_NO_PY_REGISTRATIONS=1 if `vision_amd` is imported in the same process: the reference
package brings its own fake/autograd/autocast registrations for the same schemas.
"""
import os
import types

from . import _loader


def make_overlay(dst: str, reference_pkg: str) -> str:
    """Create `<dst>/torchvision` (symlinks) and return `dst`."""
    if not os.path.isdir(reference_pkg) or not os.path.exists(os.path.join(reference_pkg, "extension.py")):
        raise FileNotFoundError(f"{reference_pkg} is not a torchvision package directory")
    for so in (_loader.KERNELS_SO, _loader.SHIM_SO, _loader.STABLE_SHIM_SO):
        if not os.path.exists(so):
            raise _loader.ExtensionMissing(f"{so} is missing; build the extension first")
    pkg = os.path.join(dst, "torchvision")
    os.makedirs(pkg, exist_ok=True)
    for name in os.listdir(reference_pkg):
        if name in ("__pycache__",) or name.startswith("_C"):
            continue
        link = os.path.join(pkg, name)
        if not os.path.lexists(link):
            os.symlink(os.path.join(reference_pkg, name), link)
    for name, target in (("_C.so", _loader.SHIM_SO), ("_C_stable.so", _loader.STABLE_SHIM_SO)):
        link = os.path.join(pkg, name)
        if os.path.lexists(link):
            os.remove(link)
        os.symlink(target, link)
    return dst


def _is_retinanet(model) -> bool:
    """True for the reference's RetinaNet (retinanet.py:324) and its subclasses: class name in the MRO, the anchor-box coder
    with unit weights (retinanet.py:433) and a head without a centerness branch."""
    if not (hasattr(model, "topk_candidates") and hasattr(model, "postprocess_detections")):
        return False
    if not any(c.__name__ == "RetinaNet" for c in type(model).__mro__):
        return False
    weights = getattr(getattr(model, "box_coder", None), "weights", None)
    if weights is None or tuple(float(w) for w in weights) != (1.0, 1.0, 1.0, 1.0):
        return False
    head = getattr(model, "head", None)
    return head is not None and hasattr(head, "classification_head") and not hasattr(getattr(head, "regression_head", None), "bbox_ctrness")


class _ImageList:
    """models/detection/image_list.py:6-24 — `tensors` + `image_sizes` is all the detectors read."""

    def __init__(self, tensors, image_sizes):
        self.tensors, self.image_sizes = tensors, image_sizes

    def to(self, device):
        return _ImageList(self.tensors.to(device), self.image_sizes)


def _tracing() -> bool:
    import torch

    return torch.jit.is_scripting() or torch.jit.is_tracing()


# ---- the fused pieces as METHODS of the reference's classes.  Each factory takes the reference's own method and returns a
# replacement with the same signature that runs the fused path when it applies (device tensors, eager mode, the configuration
# the fused launch implements) and calls the reference's method otherwise.  `fuse_detection_model` binds them to ONE model;
# `vision_amd.autofuse` installs them at class level when TVMI_AUTOFUSE=1 (the unchanged reference python, fast by default).
def make_pool_forward(orig):
    """MultiScaleRoIAlign.forward (ops/poolers.py:289-321) -> ONE multi-scale launch instead of the per-level
    where / roi_align / index_put loop (ops/poolers.py:199-222).  Scales and the level mapper are the reference's own
    (`self.scales`, `self.map_levels`, set up by ITS `_setup_scales` through `orig`'s module)."""
    import torch

    from . import poolers

    def forward(self, x, boxes, image_shapes):
        feats = [v for k, v in x.items() if k in self.featmap_names]
        if (_tracing() or len(feats) < 2 or not all(f.is_cuda for f in feats) or not len(boxes) or not boxes[0].is_cuda
                or feats[0].dtype not in (torch.float32, torch.float16, torch.bfloat16)):
            return orig(self, x, boxes, image_shapes)
        if self.scales is None or self.map_levels is None:
            self.scales, self.map_levels = poolers._setup_scales(feats, image_shapes, self.canonical_scale, self.canonical_level)
        return poolers._multiscale_roi_align(feats, boxes, tuple(self.output_size), int(self.sampling_ratio), self.scales, self.map_levels)
    return forward


def make_postprocess_detections(orig):
    """RoIHeads.postprocess_detections (roi_heads.py:680-737), batched over the images."""
    import vision_amd

    def postprocess_detections(self, class_logits, box_regression, proposals, image_shapes):
        if _tracing() or not class_logits.is_cuda:
            return orig(self, class_logits, box_regression, proposals, image_shapes)
        return vision_amd.postprocess_detections(class_logits, box_regression, proposals, image_shapes,
                                                 bbox_reg_weights=self.box_coder.weights, score_thresh=self.score_thresh,
                                                 nms_thresh=self.nms_thresh, detections_per_img=self.detections_per_img)
    return postprocess_detections


def make_filter_proposals(orig):
    """RegionProposalNetwork.filter_proposals (rpn.py:242-297), batched over images and levels."""
    import vision_amd

    def filter_proposals(self, proposals, objectness, image_shapes, num_anchors_per_level):
        if _tracing() or not proposals.is_cuda:
            return orig(self, proposals, objectness, image_shapes, num_anchors_per_level)
        return vision_amd.filter_proposals(proposals, objectness, image_shapes, num_anchors_per_level,
                                           pre_nms_top_n=self.pre_nms_top_n(), post_nms_top_n=self.post_nms_top_n(),
                                           nms_thresh=self.nms_thresh, score_thresh=self.score_thresh, min_size=self.min_size)
    return filter_proposals


def make_retinanet_postprocess(orig):
    """RetinaNet.postprocess_detections (retinanet.py:509-571); anything that is not the reference's RetinaNet arithmetic
    (a subclass with another box coder or head) keeps the reference's method."""
    import vision_amd

    def postprocess_detections(self, head_outputs, anchors, image_shapes):
        cls = head_outputs["cls_logits"]
        if _tracing() or not _is_retinanet(self) or not isinstance(cls, (list, tuple)) or not cls[0].is_cuda:
            return orig(self, head_outputs, anchors, image_shapes)
        return vision_amd.retinanet_postprocess_detections(
            cls, head_outputs["bbox_regression"], anchors, image_shapes, score_thresh=self.score_thresh,
            topk_candidates=self.topk_candidates, nms_thresh=self.nms_thresh, detections_per_img=self.detections_per_img)
    return postprocess_detections


def make_transform_forward(orig, image_list_cls=None):
    """GeneralizedRCNNTransform.forward in eval mode (transform.py:119-255): normalise + resize + pad of the batch in one launch.
    Training, targets, fixed_size / _skip_resize transforms (SSD builds its transform with fixed_size=size, ssd.py:331-334;
    transform.py:174-191) and CPU images run the reference's own forward."""
    import vision_amd

    if image_list_cls is None:
        image_list_cls = _ImageList

    def forward(self, images, targets=None):
        if (_tracing() or self.training or targets is not None or getattr(self, "fixed_size", None) is not None
                or getattr(self, "_skip_resize", False) or not len(images) or not all(im.is_cuda and im.dim() == 3 for im in images)):
            return orig(self, images, targets)
        tensors, sizes = vision_amd.transform_images(images, self.min_size, self.max_size, self.image_mean, self.image_std,
                                                     self.size_divisible)
        return image_list_cls(tensors, [tuple(s) for s in sizes]), targets
    return forward


def make_transform_postprocess(orig):
    """GeneralizedRCNNTransform.postprocess (transform.py:257-276) with the batched paste (roi_heads.py:378-500)."""
    import vision_amd

    def postprocess(self, result, image_shapes, original_image_sizes):
        if self.training:
            return result
        if _tracing() or not len(result) or not result[0]["boxes"].is_cuda:
            return orig(self, result, image_shapes, original_image_sizes)
        for i, (pred, im_s, o_im_s) in enumerate(zip(result, image_shapes, original_image_sizes)):
            boxes = vision_amd.resize_boxes(pred["boxes"], im_s, o_im_s)
            result[i]["boxes"] = boxes
            if "masks" in pred:
                result[i]["masks"] = vision_amd.paste_masks_in_image(pred["masks"], boxes, o_im_s)
            if "keypoints" in pred:
                result[i]["keypoints"] = vision_amd.resize_keypoints(pred["keypoints"], im_s, o_im_s)
        return result
    return postprocess


def make_deterministic_roi_align(orig):
    """`torchvision.ops.roi_align._roi_align` — the pure-python RoIAlign the reference detours to on GPU tensors under
    `torch.use_deterministic_algorithms(True)` because ITS backward scatters with atomics (ops/roi_align.py:276-281,
    cuda/roi_align_kernel.cu:304-327).  Ours is the tile-owner backward (roi_align_bwd.hip: one writer per pixel, fixed
    summation order, bit-reproducible), so CUDA tensors of the types the op serves stay on `torchvision::roi_align`; anything
    else (no ops loaded, other devices) keeps the reference's python."""
    import torch

    def _roi_align(input, rois, spatial_scale, pooled_height, pooled_width, sampling_ratio=-1, aligned=False):
        if (_tracing() or not input.is_cuda or not rois.is_cuda
                or input.dtype not in (torch.float32, torch.float64, torch.float16, torch.bfloat16)):
            return orig(input, rois, spatial_scale, pooled_height, pooled_width, sampling_ratio, aligned)
        return torch.ops.torchvision.roi_align(input, rois.to(input.dtype), spatial_scale, pooled_height, pooled_width,
                                               sampling_ratio, aligned)
    return _roi_align


def fuse_detection_model(model, paste_masks: bool = True, transform: bool = True):
    """Swap the fused `vision_amd` pieces into a detection model of the reference (`torchvision.models.detection`:
    Faster R-CNN / Mask R-CNN / Keypoint R-CNN = GeneralizedRCNN with RoIHeads + RPN, or RetinaNet), in place, and return it.
    The model keeps its weights, its python class and its output format; what changes is how the hot path runs:

      roi_heads.box_roi_pool / mask_roi_pool / keypoint_roi_pool  -> `vision_amd.MultiScaleRoIAlign` (ONE multi-scale launch instead of
                                                       the per-level where / roi_align / index_put loop, ops/poolers.py:199-222)
      roi_heads.postprocess_detections   -> `vision_amd.postprocess_detections`   (roi_heads.py:680-737, batched over images)
      rpn.filter_proposals               -> `vision_amd.filter_proposals`         (rpn.py:242-297)
      RetinaNet.postprocess_detections   -> `vision_amd.retinanet_postprocess_detections` (retinanet.py:509-571)
      transform.forward (eval)           -> `vision_amd.transform_images`         (transform.py:119-255: one launch per batch)
      transform.postprocess              -> the reference's method with `vision_amd.paste_masks_in_image` (roi_heads.py:378-500)

    Only attributes of THIS model are replaced (no module-level monkey patching; `vision_amd.autofuse` is the class-level,
    opt-in form of the same swaps).  Every replaced method calls the reference's own when the fused path does not apply:
    training mode / targets (targets are resized by the reference transform), CPU tensors, fixed_size / _skip_resize transforms.
    RetinaNet only: FCOS (fcos.py:426,489) and SSD / SSDlite (ssd.py:240,414) carry the same `topk_candidates` /
    `postprocess_detections` attributes but score with sqrt(cls * centerness) + BoxLinearCoder resp. softmax over one logits
    tensor — they keep the reference's post-processing.  Duck-typed: nothing of the reference package is imported."""
    import vision_amd

    def swap_pool(owner, name):
        old = getattr(owner, name, None)
        if old is None or not hasattr(old, "featmap_names") or not hasattr(old, "sampling_ratio"):
            return
        new = vision_amd.MultiScaleRoIAlign(list(old.featmap_names), tuple(old.output_size), int(old.sampling_ratio),
                                            canonical_scale=int(getattr(old, "canonical_scale", 224)),
                                            canonical_level=int(getattr(old, "canonical_level", 4)))
        setattr(owner, name, new)

    def bind(owner, name, factory):
        setattr(owner, name, types.MethodType(factory(getattr(type(owner), name)), owner))

    rh, rpn = getattr(model, "roi_heads", None), getattr(model, "rpn", None)
    if rh is not None:
        for name in ("box_roi_pool", "mask_roi_pool", "keypoint_roi_pool"):
            swap_pool(rh, name)
        if hasattr(rh, "box_coder") and hasattr(rh, "detections_per_img"):
            bind(rh, "postprocess_detections", make_postprocess_detections)
    if rpn is not None and hasattr(rpn, "pre_nms_top_n"):
        bind(rpn, "filter_proposals", make_filter_proposals)
    if rh is None and _is_retinanet(model):
        bind(model, "postprocess_detections", make_retinanet_postprocess)
    tr = getattr(model, "transform", None)
    if tr is not None and transform and hasattr(tr, "image_mean"):
        bind(tr, "forward", make_transform_forward)
    if tr is not None and paste_masks and hasattr(tr, "postprocess"):
        bind(tr, "postprocess", make_transform_postprocess)
    return model
