"""deform_conv2d config 4 depthwise forward (groups = 256): the packed kernel (dcn.dw_variant 1) against the round-2..4 kernel
(0) — time and max abs difference, fp32 / bf16 / fp16, with and without mask.   python tools/dw_variants.py [out.json]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, vision_amd
dev = "cuda"; g = torch.Generator().manual_seed(0)
B, C, H, W = 2, 256, 100, 136
sets = []
for i in range(3):
    sets.append(dict(x=torch.randn(B, C, H, W, generator=g).to(dev), off=torch.randn(B, 18, H, W, generator=g).to(dev),
                     m=torch.rand(B, 9, H, W, generator=g).to(dev), w=(torch.randn(C, 1, 3, 3, generator=g) * 0.2).to(dev),
                     b=torch.randn(C, generator=g).to(dev)))
def tm(fn, n=30):
    for i in range(5): fn(i)
    torch.cuda.synchronize(); ts = []
    for r in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(n): fn(i)
        b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) / n)
    return round(min(ts), 4), round(sorted(ts)[len(ts) // 2], 4)
out = {}
for dt in (torch.float32, torch.bfloat16, torch.float16):
    ss = [{k: v.to(dt) for k, v in s.items()} for s in sets]
    for use_mask in (False, True):
        res = {}
        for v in (0, 1):
            torch.ops.tvmi.set_option("dcn.dw_variant", v)
            call = lambda i: vision_amd.deform_conv2d(ss[i % 3]["x"], ss[i % 3]["off"], ss[i % 3]["w"], ss[i % 3]["b"], padding=1,
                                                      mask=ss[i % 3]["m"] if use_mask else None)
            res[v] = (call(0).float(), tm(call))
        key = f"{str(dt)[6:]} mask={use_mask}"
        out[key] = {"old_ms(min,med)": res[0][1], "packed_ms(min,med)": res[1][1], "max_abs_diff": float((res[0][0] - res[1][0]).abs().max()),
                    "out_absmax": float(res[0][0].abs().max())}
        print(key, out[key], flush=True)
torch.ops.tvmi.set_option("dcn.dw_variant", 1)
if len(sys.argv) > 1: json.dump(out, open(sys.argv[1], "w"), indent=1)
