"""Runs one hot-path kernel a few times (for rocprofv3 --pmc / --kernel-trace passes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, vision_amd, bench
if os.environ.get("TVMI_TOOL_SERIALIZED_PROFILER"):
    # rocprofv3 --pmc runs ONE kernel at a time: the polling push kernel of the large-NMS hand-offs would wait for a resolver
    # that cannot start.  Counter passes use the stream-event form of the same pipeline.
    torch.ops.tvmi.set_option("nms.device_handoff", 0)
for kv in filter(None, os.environ.get("TVMI_SET_OPTIONS", "").split(",")):     # e.g. TVMI_SET_OPTIONS=roi_align.order=0,nms.device_handoff=0
    name, val = kv.split("=")
    torch.ops.tvmi.set_option(name, int(val))
which = sys.argv[1] if len(sys.argv) > 1 else "roi7"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
feats, boxes, scores = bench.make_inputs(dev, 1000)
shapes = [(800, 1344)] * 4
with torch.no_grad():
    if which in ("roi7", "roi14"):
        pool = vision_amd.MultiScaleRoIAlign(["0", "1", "2", "3"], 7 if which == "roi7" else 14, 2)
        for _ in range(reps):
            pool(feats, boxes, shapes)
    elif which == "step7":   # the detector step as ONE launch: the NMS workgroups in front of the 7 x 7 RoIAlign grid (bench.py's step)
        pool = vision_amd.MultiScaleRoIAlign(["0", "1", "2", "3"], 7, 2)
        ab, asc = torch.cat(boxes), torch.cat(scores)
        img = torch.arange(4, device=dev).repeat_interleave(1000)
        for _ in range(reps):
            pool.forward_with_nms_step(feats, boxes, shapes, ab, asc, img, 0.5, 4, img, 4, 100)
    elif which == "roi7cl":
        pool = vision_amd.MultiScaleRoIAlign(["0", "1", "2", "3"], 7, 2)
        cl = {k: v.contiguous(memory_format=torch.channels_last) for k, v in feats.items()}
        for _ in range(reps):
            pool(cl, boxes, shapes)
    elif which in ("bwd7", "bwd14"):
        from vision_amd.poolers import _convert_to_roi_format
        P = 7 if which == "bwd7" else 14
        rois = _convert_to_roi_format(boxes).float()
        fl = [feats[str(i)] for i in range(4)]
        gall = torch.randn(4000, 256, P, P, device=dev)
        for _ in range(reps):
            torch.ops.tvmi.multiscale_roi_align_backward(gall, rois, [f.shape[2] for f in fl], [f.shape[3] for f in fl],
                                                         [1 / s for s in bench.STRIDES], 4, P, P, 2, False, 2, 5, 224.0, 4.0, 1e-6)
    elif which == "roipool":
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from helpers import rois_for
        g = torch.Generator().manual_seed(3)
        x = torch.randn(4, 256, 100, 168, generator=g).to(dev)
        rois = rois_for(4, 4000, 1344, 800, 32, 400, g)
        rois = rois[torch.argsort(rois[:, 0], stable=True)].to(dev)
        for _ in range(reps):
            torch.ops.torchvision.roi_pool(x, rois, 0.125, 7, 7)
    elif which.startswith("resize_"):       # resize_bilinear | resize_bilinear_aa | resize_bicubic | resize_bicubic_aa
        import torch.nn.functional as F
        g = torch.Generator().manual_seed(1)
        img = torch.rand(8, 3, 1080, 1920, generator=g).to(dev)
        mode = "bicubic" if "bicubic" in which else "bilinear"
        for _ in range(reps):
            vision_amd.interpolate(img, size=(800, 1422), mode=mode, align_corners=False, antialias=which.endswith("_aa"))
    elif which in ("nms100k", "nms100k_dense"):
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from helpers import random_boxes
        g = torch.Generator().manual_seed(7)
        canvas = 200 if which.endswith("dense") else 1000
        b = random_boxes(100_000, canvas, canvas, 1, 101, g).to(dev); s = torch.rand(100_000, generator=g).to(dev)
        for _ in range(reps):
            torch.ops.torchvision.nms(b, s, 0.5)
    elif which == "nms":
        b, s = torch.cat(boxes), torch.cat(scores)
        idx = torch.cat([torch.full((1000,), i, device=dev, dtype=torch.int64) for i in range(4)])
        for _ in range(reps):
            vision_amd.batched_nms(b, s, idx, 0.5)
    elif which in ("dcn_dw", "dcn_bf16"):
        g = torch.Generator().manual_seed(0)
        dt = torch.bfloat16 if which == "dcn_bf16" else torch.float32
        groups = 256 if which == "dcn_dw" else 1
        x = torch.randn(2, 256, 100, 136, generator=g).to(dev).to(dt)
        w = (torch.randn(256, 256 // groups, 3, 3, generator=g) * 0.01).to(dev).to(dt)
        off = torch.randn(2, 18, 100, 136, generator=g).to(dev).to(dt)
        for _ in range(reps):
            vision_amd.deform_conv2d(x, off, w, padding=1)
    elif which in ("dcn_bwd", "dcn_bwd_dw", "dcn_bwd_bf16"):
        g = torch.Generator().manual_seed(0)
        groups = 256 if which == "dcn_bwd_dw" else 1
        dt = torch.bfloat16 if which.endswith("bf16") else torch.float32
        ts = [torch.randn(2, 256, 100, 136, generator=g), torch.randn(2, 256, 100, 136, generator=g),
              torch.randn(256, 256 // groups, 3, 3, generator=g) * 0.01, torch.randn(2, 18, 100, 136, generator=g),
              torch.rand(2, 9, 100, 136, generator=g), torch.randn(256, generator=g)]
        ts = [t.to(dev).to(dt) for t in ts]
        for _ in range(reps):
            torch.ops.torchvision._deform_conv2d_backward(*ts, 1, 1, 1, 1, 1, 1, groups, 1, True)
    elif which == "dcn":
        g = torch.Generator().manual_seed(0)
        x = torch.randn(2, 256, 100, 136, generator=g).to(dev); w = (torch.randn(256, 256, 3, 3, generator=g) * 0.01).to(dev)
        off = torch.randn(2, 18, 100, 136, generator=g).to(dev)
        for _ in range(reps):
            vision_amd.deform_conv2d(x, off, w, padding=1)
torch.cuda.synchronize()
