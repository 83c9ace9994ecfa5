"""NMS / batched_nms timings (config 1 / 2 / 3 sizes), segment-major path vs global-order path."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, vision_amd
dev = torch.device("cuda:0")
def tm(fn, n=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
def boxes(n, g, span=1000.0):
    xy = torch.rand(n, 2, generator=g) * span; wh = 8 + torch.rand(n, 2, generator=g) * 120
    return torch.cat([xy, xy + wh], 1)
g = torch.Generator().manual_seed(0)
for n, S in ((4000, 4), (4000, 80), (20000, 80), (100000, 80), (100000, 1)):
    b = boxes(n, g).to(dev); s = torch.rand(n, generator=g).to(dev); idx = torch.randint(0, S, (n,), generator=g).to(dev)
    k = torch.ops.tvmi.nms_segmented(b, s, idx, 0.5)
    t = tm(lambda: torch.ops.tvmi.nms_segmented(b, s, idx, 0.5))
    print(f"batched_nms n={n} segments={S}: {t:.3f} ms  kept {len(k)}  (TVMI_NMS_SEGMAJOR={os.environ.get('TVMI_NMS_SEGMAJOR', '1')})")
for n in (1000, 4000, 100000):
    b = boxes(n, g).to(dev); s = torch.rand(n, generator=g).to(dev)
    t = tm(lambda: torch.ops.torchvision.nms(b, s, 0.5))
    print(f"nms n={n}: {t:.3f} ms")
