"""§8f timings: batched paste_masks vs the reference's python loop run on the same GPU (torch ops)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.nn.functional as F, vision_amd
from vision_amd.masks import expand_boxes, expand_masks
dev = torch.device("cuda:0")
def tm(fn, n=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
def loop_paste(masks, boxes, im_h, im_w, padding=1):
    # the reference algorithm (roi_heads.py:416-437,486-500) with torch ops on the device
    masks, scale = expand_masks(masks, padding)
    boxes = expand_boxes(boxes, scale).to(torch.int64).tolist()
    res = []
    for m, b in zip(masks, boxes):
        w = max(b[2] - b[0] + 1, 1); h = max(b[3] - b[1] + 1, 1)
        mm = F.interpolate(m[None], size=(h, w), mode="bilinear", align_corners=False)[0][0]
        im = torch.zeros((im_h, im_w), dtype=mm.dtype, device=mm.device)
        x0, x1, y0, y1 = max(b[0], 0), min(b[2] + 1, im_w), max(b[1], 0), min(b[3] + 1, im_h)
        im[y0:y1, x0:x1] = mm[(y0 - b[1]):(y1 - b[1]), (x0 - b[0]):(x1 - b[0])]
        res.append(im)
    return torch.stack(res)[:, None]
g = torch.Generator().manual_seed(0)
im_h, im_w, n = 800, 1333, 100
xy = torch.rand(n, 2, generator=g) * torch.tensor([im_w - 200.0, im_h - 200.0]); wh = 20 + torch.rand(n, 2, generator=g) * 300
boxes = torch.cat([xy, torch.minimum(xy + wh, torch.tensor([float(im_w), float(im_h)]))], 1).to(dev)
masks = torch.rand(n, 1, 28, 28, generator=g).to(dev)
a = vision_amd.paste_masks_in_image(masks, boxes, (im_h, im_w)); b = loop_paste(masks, boxes, im_h, im_w)
print("max |fused - torch loop| =", (a - b).abs().max().item(), "support equal:", bool(((a != 0) == (b != 0)).all()))
t1 = tm(lambda: vision_amd.paste_masks_in_image(masks, boxes, (im_h, im_w)))
t2 = tm(lambda: loop_paste(masks, boxes, im_h, im_w), n=3, warm=1)
byts = n * im_h * im_w * 4
print(f"paste_masks 100 x 28x28 -> 800x1333: fused {t1:.4f} ms ({byts / t1 / 1e6:.0f} GB/s written) | torch loop {t2:.2f} ms | x{t2 / t1:.0f}")

# ---- fused post-processing vs the reference's per-image torch-op chain on the same GPU
import math
from vision_amd import boxes as VB
def ref_postprocess(class_logits, box_regression, proposals, image_shapes, score_thresh=0.05, nms_thresh=0.5, dets=100):
    # roi_heads.py:680-737 with torch ops (BoxCoder.decode_single inlined), NMS = our batched_nms
    C = class_logits.shape[-1]
    boxes = torch.cat(proposals)
    w = boxes[:, 2] - boxes[:, 0]; h = boxes[:, 3] - boxes[:, 1]; cx = boxes[:, 0] + 0.5 * w; cy = boxes[:, 1] + 0.5 * h
    dx = box_regression[:, 0::4] / 10; dy = box_regression[:, 1::4] / 10
    dw = torch.clamp(box_regression[:, 2::4] / 5, max=math.log(1000 / 16)); dh = torch.clamp(box_regression[:, 3::4] / 5, max=math.log(1000 / 16))
    pcx = dx * w[:, None] + cx[:, None]; pcy = dy * h[:, None] + cy[:, None]; pw = torch.exp(dw) * w[:, None]; ph = torch.exp(dh) * h[:, None]
    pred = torch.stack((pcx - 0.5 * pw, pcy - 0.5 * ph, pcx + 0.5 * pw, pcy + 0.5 * ph), dim=2)
    scores = F.softmax(class_logits, -1)
    out = []
    n = [len(p) for p in proposals]
    for b, s, shape in zip(pred.split(n), scores.split(n), image_shapes):
        b = VB.clip_boxes_to_image(b, shape)
        lab = torch.arange(C, device=b.device).view(1, -1).expand_as(s)
        b = b[:, 1:].reshape(-1, 4); s = s[:, 1:].reshape(-1); lab = lab[:, 1:].reshape(-1)
        i = torch.where(s > score_thresh)[0]; b, s, lab = b[i], s[i], lab[i]
        k = VB.remove_small_boxes(b, 1e-2); b, s, lab = b[k], s[k], lab[k]
        k = VB.batched_nms(b, s, lab, nms_thresh)[:dets]
        out.append((b[k], s[k], lab[k]))
    return out
g = torch.Generator().manual_seed(1)
shapes = [(800, 1333)] * 4
props = [torch.cat([xy := torch.rand(1000, 2, generator=g) * 700, xy + 16 + torch.rand(1000, 2, generator=g) * 300], 1).to(dev) for _ in range(4)]
logits = (torch.randn(4000, 91, generator=g) * 3).to(dev); reg = (torch.randn(4000, 364, generator=g) * 0.5).to(dev)
t1 = tm(lambda: vision_amd.postprocess_detections(logits, reg, props, shapes, padded=True))
t2 = tm(lambda: ref_postprocess(logits, reg, props, shapes), n=5, warm=1)
print(f"postprocess_detections 4 x 1000 x 91: fused {t1:.3f} ms | per-image torch chain (with our NMS) {t2:.3f} ms | x{t2 / t1:.1f}")

# ---- fused input transform vs the reference's per-image op chain on the same GPU
def ref_transform(images, min_size=800, max_size=1333, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
    outs = []
    for im in images:
        m = torch.as_tensor(mean, dtype=im.dtype, device=im.device); s = torch.as_tensor(std, dtype=im.dtype, device=im.device)
        x = (im - m[:, None, None]) / s[:, None, None]
        h, w = x.shape[-2:]
        sc = min(min_size / min(h, w), max_size / max(h, w))
        outs.append(F.interpolate(x[None], scale_factor=sc, mode="bilinear", recompute_scale_factor=True, align_corners=False)[0])
    hp = int(math.ceil(max(o.shape[1] for o in outs) / 32) * 32); wp = int(math.ceil(max(o.shape[2] for o in outs) / 32) * 32)
    b = outs[0].new_full((len(outs), 3, hp, wp), 0)
    for i, o in enumerate(outs):
        b[i, :, : o.shape[1], : o.shape[2]].copy_(o)
    return b
imgs = [torch.rand(3, h, w, generator=g).to(dev) for h, w in ((480, 640), (720, 1280), (600, 800), (1080, 1920))]
a, _ = vision_amd.transform_images(imgs); b = ref_transform(imgs)
print("transform max |fused - torch chain| =", (a - b).abs().max().item(), tuple(a.shape))
t1 = tm(lambda: vision_amd.transform_images(imgs)); t2 = tm(lambda: ref_transform(imgs), n=5, warm=1)
print(f"GeneralizedRCNNTransform 4 images -> {tuple(a.shape)}: fused {t1:.3f} ms | per-image torch chain {t2:.3f} ms | x{t2 / t1:.1f}")
