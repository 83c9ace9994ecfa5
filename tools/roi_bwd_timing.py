"""config 2 backward legs (fused multi-scale tile-owner backward): median / min ms per call, 7x7 and 14x14, fp32 and bf16,
3 rotated input sets (as bench.py's `configs` block does).   python tools/roi_bwd_timing.py [out.json]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, vision_amd, bench
from vision_amd.poolers import _convert_to_roi_format
dev = torch.device("cuda:0")
def med(fn, n=24, warm=3):
    for i in range(warm): fn(i)
    torch.cuda.synchronize(); ts = []
    for i in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(i); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ts.sort(); return round(ts[len(ts) // 2], 4), round(ts[0], 4)
c2 = []
for i in range(3):
    f, b, _ = bench.make_inputs(dev, 300 + i)
    c2.append(([f[str(l)] for l in range(4)], _convert_to_roi_format(b).float()))
hs, ws = [t.shape[2] for t in c2[0][0]], [t.shape[3] for t in c2[0][0]]
scales, ms_args = [1.0 / st for st in bench.STRIDES], (2, 5, 224.0, 4.0, 1e-6)
out = {}
for P in (7, 14):
    for dt, tag in ((torch.float32, "fp32"), (torch.bfloat16, "bf16")):
        sets = [(r, torch.randn(4000, 256, P, P, device=dev).to(dt)) for _, r in c2]
        out[f"bwd_{P}x{P}_{tag}"] = med(lambda i: torch.ops.tvmi.multiscale_roi_align_backward(sets[i % 3][1], sets[i % 3][0], hs, ws, scales, 4, P, P, 2, False, *ms_args))
        print(f"bwd_{P}x{P}_{tag}", out[f"bwd_{P}x{P}_{tag}"], flush=True)
        del sets
if len(sys.argv) > 1: json.dump(out, open(sys.argv[1], "w"), indent=1)
