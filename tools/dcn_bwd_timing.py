"""config 4 backward (all five gradients, mask in use), groups = 1: fp32 / bf16 / fp16 median ms, and the 16-bit results against
the fp32 backward of the same (rounded) inputs.   python tools/dcn_bwd_timing.py [out.json]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, vision_amd
dev = "cuda"; g = torch.Generator().manual_seed(0)
B, C, H, W, OC = 2, 256, 100, 136, 256
ts32 = [torch.randn(B, OC, H, W, generator=g), torch.randn(B, C, H, W, generator=g), torch.randn(OC, C, 3, 3, generator=g) * 0.01,
        torch.randn(B, 18, H, W, generator=g), torch.rand(B, 9, H, W, generator=g), torch.randn(OC, generator=g)]
def med(fn, n=12):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); t.append(a.elapsed_time(b))
    t.sort(); return round(t[len(t) // 2], 4)
out = {}
tv = torch.ops.torchvision
for dt in (torch.float32, torch.bfloat16, torch.float16):
    ts = [t.to(dt).to(dev) for t in ts32]
    got = tv._deform_conv2d_backward(*ts, 1, 1, 1, 1, 1, 1, 1, 1, True)
    key = str(dt)[6:]
    out[key] = {"ms": med(lambda: tv._deform_conv2d_backward(*ts, 1, 1, 1, 1, 1, 1, 1, 1, True))}
    if dt != torch.float32:
        ref = tv._deform_conv2d_backward(*[t.float() for t in ts], 1, 1, 1, 1, 1, 1, 1, 1, True)   # fp32 backward of the rounded inputs
        names = ("grad_input", "grad_weight", "grad_offset", "grad_mask", "grad_bias")
        out[key]["max_err_over_scale"] = {n: round(float((a.float() - b).abs().max() / b.abs().max()), 5) for n, a, b in zip(names, got, ref)}
    print(key, out[key], flush=True)
if len(sys.argv) > 1: json.dump(out, open(sys.argv[1], "w"), indent=1)
