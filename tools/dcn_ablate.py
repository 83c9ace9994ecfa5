"""deform_conv2d config 4: the fused kernels with the gathers / the MFMAs switched off (dcn.ablate) — which leg bounds them?"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, vision_amd
dev = "cuda"; g = torch.Generator().manual_seed(0)
B, C, H, W = 2, 256, 100, 136
x = torch.randn(B, C, H, W, generator=g).to(dev); off = torch.randn(B, 18, H, W, generator=g).to(dev)
w = (torch.randn(256, C, 3, 3, generator=g) * 0.01).to(dev); bias = torch.randn(256, generator=g).to(dev)
def tm(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): fn()
        b.record(); torch.cuda.synchronize(); best = min(best, a.elapsed_time(b) / n)
    return best
out = {}
for dt in (torch.float32, torch.bfloat16):
    xs, os_, ws, bs = x.to(dt), off.to(dt), w.to(dt), bias.to(dt)
    for ab, name in ((0, "full"), (1, "no gathers"), (2, "no MFMAs"), (3, "neither (weights + commit + barriers)")):
        torch.ops.tvmi.set_option("dcn.ablate", ab)
        out[f"{str(dt)[6:]} {name}"] = round(tm(lambda: vision_amd.deform_conv2d(xs, os_, ws, bs, padding=1)), 4)
        print(str(dt)[6:], name, out[f"{str(dt)[6:]} {name}"], flush=True)
torch.ops.tvmi.set_option("dcn.ablate", 0)
if len(sys.argv) > 1: json.dump(out, open(sys.argv[1], "w"), indent=1)
