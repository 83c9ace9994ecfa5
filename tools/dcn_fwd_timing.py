"""deform_conv2d forward at BASELINE config 4 (2x256x100x136, k3, 256 -> 256, fp32): ms per call for the option values given on
the command line, e.g. `python tools/dcn_fwd_timing.py dcn.f32_depth2=0 dcn.f32_depth2=1`; also checks the two give the same bits."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vision_amd  # noqa: E402

dev = torch.device("cuda")
g = torch.Generator().manual_seed(100)
B = int(os.environ.get("DCN_B", "2"))
sets = []
for i in range(3):
    sets.append(dict(x=torch.randn(B, 256, 100, 136, generator=g).to(dev), off=torch.randn(B, 18, 100, 136, generator=g).to(dev),
                     w=(torch.randn(256, 256, 3, 3, generator=g) * 0.01).to(dev), b=torch.randn(256, generator=g).to(dev)))


def med(fn, n=30, warm=4):
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    ts = []
    for i in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn(i)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2], ts[0]


outs = []
for kv in sys.argv[1:] or ["dcn.f32_depth2=1"]:
    name, val = kv.split("=")
    torch.ops.tvmi.set_option(name, int(val))
    run = lambda i: vision_amd.deform_conv2d(sets[i % 3]["x"], sets[i % 3]["off"], sets[i % 3]["w"], sets[i % 3]["b"], padding=1)
    m, mn = med(run)
    outs.append(run(0).clone())
    print(f"{kv}: median {m:.4f} ms  min {mn:.4f} ms   {2 * B * 256 * 256 * 9 * 100 * 136 / m / 1e9:.1f} TFLOP/s")
if len(outs) > 1:
    print("bit-identical:", all(torch.equal(outs[0], o) for o in outs[1:]), " max diff", max(float((outs[0] - o).abs().max()) for o in outs[1:]))
