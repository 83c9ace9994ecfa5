"""deform_conv2d config 4: time the tile-shape / pipeline variants behind the `dcn.cl_variant` (16-bit channels-last kernels)
option and the XCD tile dealing (`dcn.xcd_tiles`), each checked bit for bit against variant 0 (same
arithmetic, same K order).  python tools/dcn_variants.py out.json"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, vision_amd
dev = "cuda"; g = torch.Generator().manual_seed(0)
B, C, H, W = 2, 256, 100, 136
x = torch.randn(B, C, H, W, generator=g).to(dev); off = torch.randn(B, 18, H, W, generator=g).to(dev)
msk = torch.rand(B, 9, H, W, generator=g).to(dev)
bias = torch.randn(256, generator=g).to(dev)
wfull = (torch.randn(256, C, 3, 3, generator=g) * 0.01).to(dev)
def tm(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); best = 1e9
    for _ in range(4):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): fn()
        b.record(); torch.cuda.synchronize(); best = min(best, a.elapsed_time(b) / n)
    return best
out = {}
def sweep(tag, opt, variants, dt, oc):
    xs, os_, ws, bs, ms = x.to(dt), off.to(dt), wfull[:oc].to(dt).contiguous(), bias[:oc].to(dt), msk.to(dt)
    torch.ops.tvmi.set_option(opt, 0)
    ref = vision_amd.deform_conv2d(xs, os_, ws, bs, padding=1); refm = vision_amd.deform_conv2d(xs, os_, ws, bs, padding=1, mask=ms)
    for v in variants:
        for xcd in (1, 0):
            torch.ops.tvmi.set_option(opt, v); torch.ops.tvmi.set_option("dcn.xcd_tiles", xcd)
            o = vision_amd.deform_conv2d(xs, os_, ws, bs, padding=1); om = vision_amd.deform_conv2d(xs, os_, ws, bs, padding=1, mask=ms)
            same = bool(torch.equal(o, ref)) and bool(torch.equal(om, refm))
            t = round(tm(lambda: vision_amd.deform_conv2d(xs, os_, ws, bs, padding=1)), 4)
            out[f"{tag} OC={oc} v{v} xcd_tiles={xcd}"] = {"ms": t, "bit_identical_to_v0": same}
            print(tag, oc, v, xcd, t, same, flush=True)
    torch.ops.tvmi.set_option("dcn.xcd_tiles", 1)
sweep("bf16", "dcn.cl_variant", [0, 1, 2], torch.bfloat16, 256)
sweep("f16", "dcn.cl_variant", [0, 1, 2], torch.float16, 256)
sweep("bf16", "dcn.cl_variant", [0, 1], torch.bfloat16, 128)
torch.ops.tvmi.set_option("dcn.cl_variant", 1)
if len(sys.argv) > 1: json.dump(out, open(sys.argv[1], "w"), indent=1)
