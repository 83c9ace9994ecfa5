"""Stress of the one-launch step kernel (tvmi::nms_step) — its workgroups hand over through polled memory words, so what must
never happen is a lost hand-over (wrong list), a timeout (num = -1) or a hang.  Runs the bench step's NMS on a side stream under
the RoIAlign launch and on the launch stream, thousands of times, over rotating inputs of different shapes, with random host-side
jitter, and checks EVERY result against the first (device-side compare, one host read per batch of calls).
    python tools/step_stress.py [seconds]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import vision_amd  # noqa: E402
from vision_amd import sharding  # noqa: E402

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
cases = []
for (n, S, B) in ((4000, 4, 4), (4096, 64, 16), (1000, 1, 1), (2500, 5, 5), (300, 3, 3), (4096, 4, 4)):
    xy = torch.rand(n, 2, generator=g) * 900
    wh = 10 + torch.rand(n, 2, generator=g) * 200
    b = torch.cat([xy, xy + wh], 1).to(dev)
    s = torch.rand(n, generator=g).to(dev)
    seg = (torch.arange(n) * S // n)[torch.randperm(n, generator=g)].to(dev)
    img = (seg % B).contiguous()
    ref = torch.ops.tvmi.nms_step(b, s, seg, 0.5, S, img, None, B, 100)
    torch.cuda.synchronize()
    cases.append((b, s, seg, img, S, B, [t.clone() for t in ref]))
feats, boxes, _ = bench.make_inputs(dev, 1)
pool = vision_amd.MultiScaleRoIAlign(["0", "1", "2", "3"], 7, 2)
# reference of the pooled output: the two-launch form (pre-pass launch + main launch).  Every launch below takes the folded form
# (roi_align.fold_order: the pre-pass as a workgroup of the launch, the units wait for their order entries) and is compared with it.
torch.ops.tvmi.set_option("roi_align.fold_order", 0)
with torch.no_grad():
    pooled_ref = pool(feats, boxes, [(bench.IMG_H, bench.IMG_W)] * 4).clone()
torch.ops.tvmi.set_option("roi_align.fold_order", 1)
torch.cuda.synchronize()
side = torch.cuda.Stream()
bad = torch.zeros(1, dtype=torch.int64, device=dev)
t0, calls, one_launch, folded = time.time(), 0, 0, 0
while time.time() - t0 < secs:
    for it in range(50):
        b, s, seg, img, S, B, ref = cases[(calls + it) % len(cases)]
        two = it % 3 != 0
        cur = torch.cuda.current_stream()
        st = side if two else cur
        if two:
            vision_amd.streams.wait_stream(side, cur)
        with torch.cuda.stream(st):
            k, n, p = torch.ops.tvmi.nms_step(b, s, seg, 0.5, S, img, None, B, 100)
            bad += (~torch.equal(k, ref[0]) if False else (k != ref[0]).any().to(torch.int64)) + (n != ref[1]).any().to(torch.int64) + (p != ref[2]).any().to(torch.int64)
        if it % 2 == 0:
            with torch.no_grad():
                po = pool(feats, boxes, [(bench.IMG_H, bench.IMG_W)] * 4)
            if it % 10 == 0:
                bad += (po != pooled_ref).any().to(torch.int64)
                folded += 1
        elif it % 5 == 0:   # the one-launch step: the NMS workgroups in front of the RoIAlign grid (round 6)
            with torch.no_grad():
                po, k2, n2, p2 = pool.forward_with_nms_step(feats, boxes, [(bench.IMG_H, bench.IMG_W)] * 4, b, s, seg, 0.5, S, img, B, 100)
            bad += (k2 != ref[0]).any().to(torch.int64) + (n2 != ref[1]).any().to(torch.int64) + (p2 != ref[2]).any().to(torch.int64)
            bad += (po != pooled_ref).any().to(torch.int64)
            one_launch += 1
        if two:
            vision_amd.streams.wait_stream(cur, side)
            for t in (k, n, p):
                t.record_stream(cur)
        if it % 7 == 0:
            time.sleep(0.0002)
    calls += 50
    nbad = int(bad.item())
    assert nbad == 0, f"{nbad} mismatching results after {calls} calls"
print(f"step_stress: {one_launch} one-launch steps (RoIAlign output and NMS results compared) and {folded} compared plain folded launches among them;", end=" ")
print(f"step_stress: {calls} calls of tvmi::nms_step in {time.time() - t0:.1f} s (one and two streams, under RoIAlign launches): all results identical to the first")
