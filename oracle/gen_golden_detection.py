#!/usr/bin/env python3
"""oracle/gen_golden_detection.py — TEST INFRASTRUCTURE ONLY.

Golden vectors for the python-level detection glue that SURVEY.md §8f lists as "next"
(mask pasting, proposal / detection post-processing).  They are produced by importing the
REAL reference python package (/root/reference/torchvision, through the symlink overlay of
vision_amd.integration so that `torchvision.ops.*` resolve to the reference's own CPU kernels
from oracle/_ref) and calling its functions on seeded CPU inputs.  Runs only where
/root/reference exists; the vectors travel with the repository (tests/golden/detection.npz).
"""
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("TVMI_NO_PY_REGISTRATIONS", "1")  # the reference package registers its own
from oracle import build_ref  # noqa: E402
from oracle import oracle as O  # noqa: E402
from oracle.gen_golden import save  # noqa: E402


def import_reference():
    from vision_amd import integration

    build_ref.build()
    assert O.load_reference()
    dst = tempfile.mkdtemp(prefix="tv_overlay_")
    sys.path.insert(0, integration.make_overlay(dst, "/root/reference/torchvision"))
    import torchvision  # noqa: F401  (the unmodified reference python)

    return torchvision


def main():
    if not build_ref.have_reference():
        raise SystemExit("gen_golden_detection.py needs /root/reference")
    import_reference()
    from torchvision.models.detection.roi_heads import paste_masks_in_image

    d = {}
    # ------------------------------------------------------------ paste_masks_in_image
    g = torch.Generator().manual_seed(21)
    im_h, im_w = 67, 83
    n = 14
    xy = torch.rand(n, 2, generator=g) * torch.tensor([im_w - 10.0, im_h - 10.0])
    wh = 1 + torch.rand(n, 2, generator=g) * torch.tensor([50.0, 40.0])
    boxes = torch.cat([xy, torch.minimum(xy + wh, torch.tensor([float(im_w), float(im_h)]))], 1)
    boxes[0] = torch.tensor([0.0, 0.0, float(im_w), float(im_h)])      # whole image (expands past every edge)
    boxes[1] = torch.tensor([10.3, 12.7, 10.9, 13.1])                  # sub-pixel box -> 1x1 .. 2x2 paste
    boxes[2] = torch.tensor([im_w - 3.0, im_h - 2.5, float(im_w), float(im_h)])  # bottom-right corner
    boxes[3] = torch.tensor([5.0, 5.0, 33.0, 33.0])                    # 28 px: resized size == padded mask size (M+2)
    for M, pad in ((28, 1), (14, 2), (7, 0)):
        masks = torch.rand(n, 1, M, M, generator=g)
        d[f"paste_masks_M{M}_p{pad}"] = masks
        d[f"paste_out_M{M}_p{pad}"] = paste_masks_in_image(masks, boxes, (im_h, im_w), padding=pad)
    d["paste_boxes"] = boxes
    d["paste_shape"] = np.array([im_h, im_w], dtype=np.int64)

    # ------------------------------------------------------------ RoIHeads.postprocess_detections
    from torchvision.models.detection.roi_heads import RoIHeads
    from torchvision.models.detection.rpn import RegionProposalNetwork

    g = torch.Generator().manual_seed(31)
    shapes = [(240, 320), (200, 300)]
    per_img, C = [70, 50], 7
    props = []
    for (h, w), r in zip(shapes, per_img):
        xy = torch.rand(r, 2, generator=g) * torch.tensor([w * 0.8, h * 0.8])
        wh = 4 + torch.rand(r, 2, generator=g) * torch.tensor([w * 0.5, h * 0.5])
        # clustered proposals so that the per-class NMS really suppresses
        xy[r // 2:] = xy[: r - r // 2] + torch.randn(r - r // 2, 2, generator=g) * 3
        wh[r // 2:] = wh[: r - r // 2] * (1 + 0.05 * torch.randn(r - r // 2, 2, generator=g))
        props.append(torch.cat([xy, xy + wh], 1))
    R = sum(per_img)
    logits = torch.randn(R, C, generator=g) * 2.5
    reg = torch.randn(R, 4 * C, generator=g) * 0.7
    reg[::9, 2::4] = 60.0  # exercise bbox_xform_clip
    heads = RoIHeads(None, None, None, 0.5, 0.5, 512, 0.25, None, score_thresh=0.05, nms_thresh=0.5, detections_per_img=20)
    boxes_l, scores_l, labels_l = heads.postprocess_detections(logits, reg, props, shapes)
    d.update(det_logits=logits, det_reg=reg, det_shapes=np.array(shapes, dtype=np.int64), det_per_img=np.array(per_img, dtype=np.int64))
    for i in range(len(shapes)):
        d[f"det_props{i}"], d[f"det_boxes{i}"], d[f"det_scores{i}"], d[f"det_labels{i}"] = props[i], boxes_l[i], scores_l[i], labels_l[i]
        assert 3 < len(boxes_l[i]) <= 20

    # ------------------------------------------------------------ RegionProposalNetwork.filter_proposals
    levels = [400, 100, 25]
    A = sum(levels)
    anchors = []
    for n, size in zip(levels, (32.0, 64.0, 128.0)):
        c = torch.rand(n, 2, generator=g) * torch.tensor([320.0, 240.0])
        anchors.append(torch.cat([c - size / 2, c + size / 2], 1))
    anchors = torch.cat(anchors)[None].expand(2, A, 4).contiguous()
    deltas = torch.randn(2, A, 4, generator=g) * 0.4
    objectness = torch.randn(2, A, generator=g) * 2
    rpn = RegionProposalNetwork(None, None, 0.7, 0.3, 256, 0.5, dict(training=60, testing=60), dict(training=40, testing=40),
                                nms_thresh=0.7, score_thresh=0.1).eval()
    proposals = rpn.box_coder.decode(deltas.reshape(-1, 4), [anchors[0], anchors[1]]).view(2, -1, 4)
    fb, fs = rpn.filter_proposals(proposals, objectness, shapes, levels)
    d.update(rpn_anchors=anchors, rpn_deltas=deltas, rpn_objectness=objectness, rpn_proposals=proposals,
             rpn_levels=np.array(levels, dtype=np.int64))
    for i in range(2):
        d[f"rpn_boxes{i}"], d[f"rpn_scores{i}"] = fb[i], fs[i]
        assert 5 < len(fb[i]) <= 40

    # ------------------------------------------------------------ GeneralizedRCNNTransform.forward (eval)
    from torchvision.models.detection.transform import GeneralizedRCNNTransform

    g = torch.Generator().manual_seed(41)
    imgs = [torch.rand(3, h, w, generator=g) for h, w in ((47, 83), (120, 64), (33, 33), (90, 211))]
    for tag, kw in (("a", dict(min_size=96, max_size=160)), ("b", dict(min_size=64, max_size=100)),
                    ("c", dict(min_size=50, max_size=80, fixed_size=(72, 56)))):
        tr = GeneralizedRCNNTransform(image_mean=[0.485, 0.456, 0.406], image_std=[0.229, 0.224, 0.225], **kw).eval()
        il, _ = tr([im.clone() for im in imgs])
        d[f"xform_{tag}_out"] = il.tensors
        d[f"xform_{tag}_sizes"] = np.array(il.image_sizes, dtype=np.int64)
    for i, im in enumerate(imgs):
        d[f"xform_img{i}"] = im
    save("detection", **d)


if __name__ == "__main__":
    main()
