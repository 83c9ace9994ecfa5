#!/usr/bin/env python3
"""Times the config-2 multi-scale RoIAlign forward 7x7 fp32 (the op: pre-pass + launch) with whatever libtvmi_kernels.so is in
vision_amd/_lib — used with the diagnostic variants of tools/build_variant.sh.  Prints median / min ms over 4 rotated input sets."""
import os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, vision_amd, bench
for kv in filter(None, os.environ.get("TVMI_SET_OPTIONS", "").split(",")):   # e.g. TVMI_SET_OPTIONS=roi_align.fold_order=0
    torch.ops.tvmi.set_option(kv.split("=")[0], int(kv.split("=")[1]))
dev = torch.device("cuda:0")
sets = [bench.make_inputs(dev, 1000 + 97 * i) for i in range(4)]
shapes = [(800, 1344)] * 4
P = int(sys.argv[2]) if len(sys.argv) > 2 else 7
pool = vision_amd.MultiScaleRoIAlign(["0", "1", "2", "3"], P, 2)
dt = torch.bfloat16 if len(sys.argv) > 3 and sys.argv[3] == "bf16" else torch.float32
sets = [({k: v.to(dt) for k, v in f.items()}, b) for f, b, _ in sets]
ts = []
with torch.no_grad():
    for i in range(8):
        pool(sets[i % 4][0], sets[i % 4][1], shapes)
    torch.cuda.synchronize()
    for i in range(60):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        pool(sets[i % 4][0], sets[i % 4][1], shapes)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
print(f"{sys.argv[1] if len(sys.argv) > 1 else '?':>10s} P={P} {str(dt)[6:]}: median {statistics.median(ts):.4f} ms  min {min(ts):.4f} ms")
