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

    Only attributes of THIS model are replaced (no module-level monkey patching).  Training mode keeps the reference transform
    (targets are resized there) and the reference post-processing is not used in training anyway.  Duck-typed: nothing of the
    reference package is imported here."""
    import vision_amd

    def swap_pool(owner, name):
        old = getattr(owner, name, None)
        if old is None or not hasattr(old, "featmap_names") or not hasattr(old, "sampling_ratio"):
            return
        new = vision_amd.MultiScaleRoIAlign(list(old.featmap_names), tuple(old.output_size), int(old.sampling_ratio),
                                            canonical_scale=int(getattr(old, "canonical_scale", 224)),
                                            canonical_level=int(getattr(old, "canonical_level", 4)))
        setattr(owner, name, new)

    rh, rpn = getattr(model, "roi_heads", None), getattr(model, "rpn", None)
    if rh is not None:
        for name in ("box_roi_pool", "mask_roi_pool", "keypoint_roi_pool"):
            swap_pool(rh, name)

        def postprocess_detections(self, class_logits, box_regression, proposals, image_shapes):
            return vision_amd.postprocess_detections(class_logits, box_regression, proposals, image_shapes,
                                                     bbox_reg_weights=self.box_coder.weights, score_thresh=self.score_thresh,
                                                     nms_thresh=self.nms_thresh, detections_per_img=self.detections_per_img)
        rh.postprocess_detections = types.MethodType(postprocess_detections, rh)
    if rpn is not None and hasattr(rpn, "pre_nms_top_n"):
        def filter_proposals(self, proposals, objectness, image_shapes, num_anchors_per_level):
            return vision_amd.filter_proposals(proposals, objectness, image_shapes, num_anchors_per_level,
                                               pre_nms_top_n=self.pre_nms_top_n(), post_nms_top_n=self.post_nms_top_n(),
                                               nms_thresh=self.nms_thresh, score_thresh=self.score_thresh, min_size=self.min_size)
        rpn.filter_proposals = types.MethodType(filter_proposals, rpn)
    if rh is None and hasattr(model, "topk_candidates") and hasattr(model, "postprocess_detections"):   # RetinaNet
        def retina_post(self, head_outputs, anchors, image_shapes):
            return vision_amd.retinanet_postprocess_detections(
                head_outputs["cls_logits"], head_outputs["bbox_regression"], anchors, image_shapes, score_thresh=self.score_thresh,
                topk_candidates=self.topk_candidates, nms_thresh=self.nms_thresh, detections_per_img=self.detections_per_img)
        model.postprocess_detections = types.MethodType(retina_post, model)
    tr = getattr(model, "transform", None)
    if tr is not None and transform and hasattr(tr, "image_mean"):
        ref_forward = tr.forward

        def fused_forward(images, targets=None):
            if tr.training or targets is not None:
                return ref_forward(images, targets)
            tensors, sizes = vision_amd.transform_images(images, tr.min_size, tr.max_size, tr.image_mean, tr.image_std,
                                                         tr.size_divisible)
            image_list = type("ImageList", (), {})()      # models/detection/image_list.py: `tensors` + `image_sizes`, nothing else is read
            image_list.tensors, image_list.image_sizes = tensors, [tuple(s) for s in sizes]
            return image_list, targets
        tr.forward = fused_forward
    if tr is not None and paste_masks and hasattr(tr, "postprocess"):
        def postprocess(self, result, image_shapes, original_image_sizes):
            if self.training:
                return result
            for i, (pred, im_s, o_im_s) in enumerate(zip(result, image_shapes, original_image_sizes)):
                boxes = vision_amd.resize_boxes(pred["boxes"], im_s, o_im_s)
                result[i]["boxes"] = boxes
                if "masks" in pred:
                    result[i]["masks"] = vision_amd.paste_masks_in_image(pred["masks"], boxes, o_im_s)
                if "keypoints" in pred:
                    result[i]["keypoints"] = vision_amd.resize_keypoints(pred["keypoints"], im_s, o_im_s)
            return result
        tr.postprocess = types.MethodType(postprocess, tr)
    return model
