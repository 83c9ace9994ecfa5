"""Config-2 RoIAlign backward timing only (per level), for knob sweeps."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, vision_amd, bench
from vision_amd.poolers import _convert_to_roi_format, LevelMapper
dev = torch.device("cuda:0"); tv = torch.ops.torchvision
def tm(fn, n=10, warm=3):
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
fl = [feats[str(i)] for i in range(4)]
for P in (7, 14):
    sel = [torch.nonzero(levels == l)[:, 0] for l in range(4)]
    grads = [torch.randn(len(s), 256, P, P, device=dev) for s in sel]
    tot = 0
    for l in range(4):
        f = fl[l]; rr = rois[sel[l]].contiguous()
        t = tm(lambda: tv._roi_align_backward(grads[l], rr, scales[l], P, P, f.shape[0], 256, f.shape[2], f.shape[3], 2, False))
        tz = tm(lambda: torch.zeros_like(f))
        print(f"P={P} level {l}: K={len(sel[l])} map {tuple(f.shape)} bwd {t:.4f} ms (zeros alone {tz:.4f})")
        tot += t
    print(f"P={P} total {tot:.4f} ms")
