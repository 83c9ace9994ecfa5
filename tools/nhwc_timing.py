"""channels_last multi-scale RoIAlign 7x7 (config 2 shapes): ordered / XCD-partitioned placement (roi_align.order 1) against
input order (0), the NCHW kernel beside it, and bit equality of the two layouts' outputs.   python tools/nhwc_timing.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, vision_amd, bench
from vision_amd.poolers import _convert_to_roi_format
dev = torch.device("cuda:0")
sets = []
for i in range(3):
    f, b, _ = bench.make_inputs(dev, 300 + i)
    sets.append(([f[str(l)].contiguous(memory_format=torch.channels_last) for l in range(4)], _convert_to_roi_format(b).float(), [f[str(l)] for l in range(4)]))
scales = [1.0 / s for s in bench.STRIDES]; args = (7, 7, 2, False, 2, 5, 224.0, 4.0, 1e-6)
def med(fn, n=30):
    for i in range(4): fn(i)
    torch.cuda.synchronize(); ts = []
    for i in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); a.record(); fn(i); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ts.sort(); return round(ts[len(ts) // 2], 4), round(ts[0], 4)
for order in (1, 0):
    torch.ops.tvmi.set_option("roi_align.order", order)
    print("order", order, "nhwc fp32", med(lambda i: torch.ops.tvmi.multiscale_roi_align(sets[i % 3][0], sets[i % 3][1], scales, *args)), flush=True)
torch.ops.tvmi.set_option("roi_align.order", 1)
print("nchw fp32", med(lambda i: torch.ops.tvmi.multiscale_roi_align(sets[i % 3][2], sets[i % 3][1], scales, *args)))
a = torch.ops.tvmi.multiscale_roi_align(sets[0][0], sets[0][1], scales, *args); b = torch.ops.tvmi.multiscale_roi_align(sets[0][2], sets[0][1], scales, *args)
print("nhwc == nchw bits:", torch.equal(a, b))
h = [([t.to(torch.bfloat16).contiguous(memory_format=torch.channels_last) for t in s[2]], s[1]) for s in sets]
print("nhwc bf16", med(lambda i: torch.ops.tvmi.multiscale_roi_align(h[i % 3][0], h[i % 3][1], scales, *args)))
