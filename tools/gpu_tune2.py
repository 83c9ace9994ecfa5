import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, vision_amd, bench
dev = torch.device("cuda:0"); lib = vision_amd._loader.kernels(); tv = torch.ops.torchvision
def tm(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
feats, boxes, scores = bench.make_inputs(dev, 1000)
pool7 = vision_amd.MultiScaleRoIAlign(["0", "1", "2", "3"], 7, 2); pool14 = vision_amd.MultiScaleRoIAlign(["0", "1", "2", "3"], 14, 2)
shapes = [(800, 1344)] * 4
lib.tvmi_debug_set(0, 0)
with torch.no_grad():
    ref7 = pool7(feats, boxes, shapes).clone(); ref14 = pool14(feats, boxes, shapes).clone()
    for variant, chunk, force, dma, order in eval(sys.argv[1]) if len(sys.argv) > 1 else [(0, 32, 0, 0, 0), (1, 32, 0, 0, 0), (1, 32, 0, 1, 1)]:
        lib.tvmi_debug_set(0, variant); lib.tvmi_debug_set(1, chunk); lib.tvmi_debug_set(2, force); lib.tvmi_debug_set(3, dma); lib.tvmi_debug_set(4, order)
        o7 = pool7(feats, boxes, shapes); o14 = pool14(feats, boxes, shapes)
        e7 = (o7 - ref7).abs().max().item(); e14 = (o14 - ref14).abs().max().item()
        t7 = tm(lambda: pool7(feats, boxes, shapes)); t14 = tm(lambda: pool14(feats, boxes, shapes))
        print(f"roi_align variant={variant} chunk={chunk} force={force} dma={dma} order={order}: 7x7 {t7:.4f} ms (err {e7:.1e})  14x14 {t14:.4f} ms (err {e14:.1e})", flush=True)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 256, 200, 272, generator=g).to(dev)
    xy = torch.rand(1000, 2, generator=g) * torch.tensor([1088 - 64.0, 800 - 64.0]); wh = 16 + torch.rand(1000, 2, generator=g) * 284
    r = torch.cat([torch.zeros(1000, 1), xy, torch.minimum(xy + wh, torch.tensor([1088.0, 800.0]))], 1).to(dev)
    for variant, chunk, force, dma in [(1, 32, 0, 1), (1, 16, 0, 1), (1, 32, 0, 0)]:
        lib.tvmi_debug_set(0, variant); lib.tvmi_debug_set(1, chunk); lib.tvmi_debug_set(2, force); lib.tvmi_debug_set(3, dma)
        print(f"config1 variant={variant} chunk={chunk} force={force} dma={dma}: {tm(lambda: tv.roi_align(x, r, 0.25, 7, 7, 2, False)):.4f} ms", flush=True)
