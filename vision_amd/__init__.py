"""vision_amd — MI355X (gfx950) native implementation of torchvision's custom operator hot
path (nms, roi_align, roi_pool, ps_roi_*, deform_conv2d, rotated box_iou, resize) behind the
reference's own `torchvision::` dispatcher schemas.  See DESIGN.md / INTEGRATION.md."""
import os as _os

from . import _loader

__version__ = "0.1.0"


def load(register_python: bool = True):
    """Load the native libraries and the python-level fake/autograd/autocast registrations.
    Pass register_python=False (or set TVMI_NO_PY_REGISTRATIONS=1) when the reference `torchvision`
    python package is going to be imported over this library: it brings its own
    _meta_registrations / _autograd_registrations for the `torchvision::` schemas, so only the
    `tvmi::` namespace is registered here then."""
    _loader.load()
    from . import _registrations

    _registrations.register_all(torchvision_schemas=bool(register_python))


def override_aten_upsample(enable: bool = True) -> bool:
    """Opt-in: put our resize kernels on the CUDA key of aten::upsample_{nearest,bilinear,bicubic}2d,
    _upsample_{bilinear,bicubic}2d_aa and _upsample_nearest_exact2d (functional + .out), so that every
    F.interpolate call of the unchanged reference python (models/detection/transform.py:65-72,
    ops/feature_pyramid_network.py:194, roi_heads.py:427, transforms/v2/functional/_geometry.py:344) lands in
    vision_amd/csrc/resize.hip.  Returns the previous state.  Also enabled by TVMI_OVERRIDE_ATEN_UPSAMPLE=1."""
    import torch

    _loader.assert_has_ops()
    return bool(torch.ops.tvmi.override_aten_upsample(bool(enable)))


try:
    load(register_python=_os.environ.get("TVMI_NO_PY_REGISTRATIONS", "0") != "1")
    if _os.environ.get("TVMI_OVERRIDE_ATEN_UPSAMPLE", "0") == "1":
        override_aten_upsample(True)
except _loader.ExtensionMissing:
    if _os.environ.get("TVMI_ALLOW_MISSING", "0") != "1":
        raise

from .boxes import (  # noqa: E402,F401
    batched_nms,
    box_area,
    box_iou,
    clip_boxes_to_image,
    complete_box_iou,
    distance_box_iou,
    generalized_box_iou,
    nms,
    remove_small_boxes,
)
from .roi_ops import (  # noqa: E402,F401
    PSRoIAlign,
    PSRoIPool,
    RoIAlign,
    RoIPool,
    ps_roi_align,
    ps_roi_pool,
    roi_align,
    roi_pool,
)
from .deform_conv import DeformConv2d, deform_conv2d  # noqa: E402,F401
from .poolers import LevelMapper, MultiScaleRoIAlign  # noqa: E402,F401
from .resize import interpolate, resize  # noqa: E402,F401
from .masks import expand_boxes, expand_masks, paste_masks_in_image  # noqa: E402,F401
from .detection_post import filter_proposals, postprocess_detections, retinanet_postprocess_detections  # noqa: E402,F401
from .integration import fuse_detection_model  # noqa: E402,F401
from .transform import resize_boxes, resize_keypoints, resized_size, transform_images  # noqa: E402,F401
from .transform import transform as transform_with_targets  # noqa: E402,F401
from . import sharding  # noqa: E402,F401
from . import streams  # noqa: E402,F401
