"""Kernel tuning visit: RoIAlign variants on the bench workload + NMS timings/parity."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import vision_amd
import bench
from oracle import oracle as O
O.load_reference()
dev = torch.device("cuda:0")
lib = vision_amd._loader.kernels()
tv = torch.ops.torchvision

def tm(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n

feats, boxes, scores = bench.make_inputs(dev, 1000)
pool7 = vision_amd.MultiScaleRoIAlign(["0", "1", "2", "3"], 7, 2)
pool14 = vision_amd.MultiScaleRoIAlign(["0", "1", "2", "3"], 14, 2)
shapes = [(800, 1344)] * 4
lib.tvmi_debug_set(0, 0)
with torch.no_grad():
    ref7 = pool7(feats, boxes, shapes).clone(); ref14 = pool14(feats, boxes, shapes).clone()
    for variant, chunk, force in [(0, 32, 0), (1, 32, 0), (1, 16, 0), (1, 64, 0), (1, 8, 0), (1, 32, 1), (1, 64, 1), (1, 128, 0), (1, 256, 0)]:
        lib.tvmi_debug_set(0, variant); lib.tvmi_debug_set(1, chunk); lib.tvmi_debug_set(2, force)
        o7 = pool7(feats, boxes, shapes); o14 = pool14(feats, boxes, shapes)
        e7 = (o7 - ref7).abs().max().item(); e14 = (o14 - ref14).abs().max().item()
        t7 = tm(lambda: pool7(feats, boxes, shapes)); t14 = tm(lambda: pool14(feats, boxes, shapes))
        print(f"roi_align ms variant={variant} chunk={chunk} force={force}: 7x7 {t7:.4f} ms (err {e7:.1e})  14x14 {t14:.4f} ms (err {e14:.1e})", flush=True)
    lib.tvmi_debug_set(0, 1); lib.tvmi_debug_set(1, 32); lib.tvmi_debug_set(2, 0)
    # config 1: single 200x272 map, big RoIs
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 256, 200, 272, generator=g).to(dev)
    xy = torch.rand(1000, 2, generator=g) * torch.tensor([1088 - 64.0, 800 - 64.0]); wh = 16 + torch.rand(1000, 2, generator=g) * 284
    r = torch.cat([torch.zeros(1000, 1), xy, torch.minimum(xy + wh, torch.tensor([1088.0, 800.0]))], 1).to(dev)
    for variant, chunk, force in [(0, 32, 0), (1, 32, 0), (1, 32, 1), (1, 64, 1)]:
        lib.tvmi_debug_set(0, variant); lib.tvmi_debug_set(1, chunk); lib.tvmi_debug_set(2, force)
        t = tm(lambda: tv.roi_align(x, r, 0.25, 7, 7, 2, False))
        print(f"config1 roi_align variant={variant} chunk={chunk} force={force}: {t:.4f} ms", flush=True)
    lib.tvmi_debug_set(0, 1); lib.tvmi_debug_set(1, 32); lib.tvmi_debug_set(2, 0)

# ---- NMS
g = torch.Generator().manual_seed(0)
for n in (1000, 4000, 5000, 20000, 100000):
    xy = torch.rand(n, 2, generator=g) * 900; wh = 1 + torch.rand(n, 2, generator=g) * 100
    b = torch.cat([xy, xy + wh], 1); s = torch.rand(n, generator=g)
    bg, sg = b.to(dev), s.to(dev)
    kg = tv.nms(bg, sg, 0.5)
    ok = None
    if n <= 20000:
        ok = bool(torch.equal(kg.cpu(), tv.nms(b, s, 0.5)))
    t = tm(lambda: tv.nms(bg, sg, 0.5), n=10)
    print(f"nms n={n} kept={kg.numel()} exact_vs_reference_cpu={ok} gpu_ms={t:.4f}", flush=True)
all_b = torch.cat(boxes); all_s = torch.cat(scores)
idx = torch.cat([torch.full((1000,), i, device=dev, dtype=torch.int64) for i in range(4)])
t = tm(lambda: vision_amd.batched_nms(all_b, all_s, idx, 0.5))
print(f"batched_nms 4x1000 (bench) {t:.4f} ms")
n = 100000
xy = torch.rand(n, 2, generator=g) * 900; wh = 1 + torch.rand(n, 2, generator=g) * 100
b = torch.cat([xy, xy + wh], 1).to(dev); s = torch.rand(n, generator=g).to(dev); ids = torch.randint(0, 80, (n,), generator=g).to(dev)
t = tm(lambda: vision_amd.batched_nms(b, s, ids, 0.5), n=5)
print(f"batched_nms 100k x 80 classes {t:.4f} ms kept={vision_amd.batched_nms(b, s, ids, 0.5).numel()}")
