"""BASELINE config 2 matrix: MultiScale RoIAlign fwd / bwd, 7x7 and 14x14, fp32 / bf16 / fp16 (+ config 1, 3, 4 timings)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, vision_amd, bench
from vision_amd.poolers import _convert_to_roi_format, LevelMapper
dev = torch.device("cuda:0"); tv = torch.ops.torchvision
def tm(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
feats, boxes, scores = bench.make_inputs(dev, 1000)
rois = _convert_to_roi_format(boxes); levels = LevelMapper(2, 5)(boxes)
scales = [1 / s for s in bench.STRIDES]
res = {}
for dt in (torch.float32, torch.bfloat16, torch.float16):
    fl = [feats[str(i)].to(dt) for i in range(4)]; r = rois.to(dt)
    in_bytes = sum(f.numel() * f.element_size() for f in fl)
    for P in (7, 14):
        t = tm(lambda: torch.ops.tvmi.multiscale_roi_align(fl, r, scales, P, P, 2, False, 2, 5, 224.0, 4.0, 1e-6))
        out_bytes = 4000 * 256 * P * P * fl[0].element_size()
        res[f"fwd_{P}x{P}_{str(dt)[6:]}"] = (t, (in_bytes + out_bytes) / t / 1e6)
        # backward: per level (the autograd path of the per-level loop)
        sel = [torch.nonzero(levels == l)[:, 0] for l in range(4)]
        grads = [torch.randn(len(s), 256, P, P, device=dev).to(dt) for s in sel]
        def bwd():
            for l in range(4):
                f = fl[l]
                tv._roi_align_backward(grads[l], r[sel[l]], scales[l], P, P, f.shape[0], 256, f.shape[2], f.shape[3], 2, False)
        t = tm(bwd, n=10)
        res[f"bwd_{P}x{P}_{str(dt)[6:]}"] = (t, (out_bytes + 2 * in_bytes) / t / 1e6)
        gall = torch.randn(4000, 256, P, P, device=dev).to(dt)
        hs = [f.shape[2] for f in fl]; ws = [f.shape[3] for f in fl]
        t = tm(lambda: torch.ops.tvmi.multiscale_roi_align_backward(gall, r, hs, ws, scales, 4, P, P, 2, False, 2, 5, 224.0, 4.0, 1e-6), n=10)
        res[f"bwd_fused_{P}x{P}_{str(dt)[6:]}"] = (t, (out_bytes + 2 * in_bytes) / t / 1e6)
for k, (t, gbs) in res.items(): print(f"config2 {k}: {t:.4f} ms  ({gbs:.0f} GB/s algorithmic)")
g = torch.Generator().manual_seed(0)
x = torch.randn(1, 256, 200, 272, generator=g).to(dev)
xy = torch.rand(1000, 2, generator=g) * torch.tensor([1088 - 64.0, 800 - 64.0]); wh = 16 + torch.rand(1000, 2, generator=g) * 284
r1 = torch.cat([torch.zeros(1000, 1), xy, torch.minimum(xy + wh, torch.tensor([1088.0, 800.0]))], 1).to(dev)
sc = torch.rand(1000, generator=g).to(dev)
t_roi = tm(lambda: tv.roi_align(x, r1, 0.25, 7, 7, 2, False)); t_nms = tm(lambda: tv.nms(r1[:, 1:].contiguous(), sc, 0.5))
print(f"config1 roi_align 7x7 {t_roi:.4f} ms + nms(1000) {t_nms:.4f} ms -> {1000 / (t_roi + t_nms) * 1e3:.0f} boxes/s")
