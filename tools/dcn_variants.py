"""GPU-box measurement of the workgroup-tile variants of the fused deform_conv2d kernels at config 4 (g = 1), each result
compared with the default variant's.  Usage: python tools/dcn_variants.py out.json"""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vision_amd

dev = torch.device("cuda:0")
res = {}


def tm(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


g = torch.Generator().manual_seed(0)
for dt in (torch.bfloat16, torch.float16, torch.float32):
    x = torch.randn(2, 256, 100, 136, generator=g).to(dev).to(dt)
    w = (torch.randn(256, 256, 3, 3, generator=g) * 0.01).to(dev).to(dt)
    off = torch.randn(2, 18, 100, 136, generator=g).to(dev).to(dt)
    msk = torch.rand(2, 9, 100, 136, generator=g).to(dev).to(dt)
    want = None
    for cl in (0, 1, 0, 1):
        torch.ops.tvmi.set_option("dcn.channels_last_gather", cl)
        y = vision_amd.deform_conv2d(x, off, w, padding=1, mask=msk)
        if want is None:
            want = y
        err = float((y.float() - want.float()).abs().max())
        t0 = tm(lambda: vision_amd.deform_conv2d(x, off, w, padding=1))
        t1 = tm(lambda: vision_amd.deform_conv2d(x, off, w, padding=1, mask=msk))
        key = f"dcn_g1_{str(dt)[6:]}_channels_last_gather{cl}" + ("" if f"dcn_g1_{str(dt)[6:]}_channels_last_gather{cl}" not in res else "_again")
        res[key] = dict(ms_nomask=round(t0, 4), ms_mask=round(t1, 4), max_abs_diff_vs_planar=err)
        print(key, res[key], flush=True)
    torch.ops.tvmi.set_option("dcn.channels_last_gather", 1)
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
