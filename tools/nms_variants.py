"""GPU-box measurement of the large-NMS pipeline variants (100k boxes, sparse 1000-px and dense 200-px canvas):
re-planning on the survivors off / on with different first-phase sizes and depths, every result checked against the
first variant's index list (which the parity tests compare with the reference).  Usage: python tools/nms_variants.py out.json"""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import vision_amd  # noqa: F401
from helpers import random_boxes

dev = torch.device("cuda:0")
res = {}


def tm(fn, n=10, warm=3):
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


cases = {}
for canvas in (1000, 200):
    g = torch.Generator().manual_seed(7)
    b = random_boxes(100_000, canvas, canvas, 1, 101, g).to(dev)
    s = torch.rand(100_000, generator=g).to(dev)
    cases[canvas] = (b, s)
g = torch.Generator().manual_seed(11)
cases["30k_500"] = (random_boxes(30_000, 500, 500, 1, 101, g).to(dev), torch.rand(30_000, generator=g).to(dev))
want = {}
variants = [(0, 16, 3, 0), (24576, 16, 3, 0), (24576, 16, 3, 1), (24576, 16, 3, 0), (24576, 16, 3, 1), (0, 16, 3, 1), (24576, 8, 2, 1), (16384, 16, 4, 1)]
for min_boxes, divisor, max_replans, handoff in variants:
    torch.ops.tvmi.set_option("nms.replan_min_boxes", min_boxes)
    torch.ops.tvmi.set_option("nms.replan_divisor", divisor)
    torch.ops.tvmi.set_option("nms.replan_max", max_replans)
    torch.ops.tvmi.set_option("nms.device_handoff", handoff)
    for name, (b, s) in cases.items():
        keep = torch.ops.torchvision.nms(b, s, 0.5)
        if name not in want:
            want[name] = keep
        same = bool(keep.numel() == want[name].numel() and torch.equal(keep, want[name]))
        t = tm(lambda: torch.ops.torchvision.nms(b, s, 0.5))
        key = f"nms_{name}_replan_min{min_boxes}_div{divisor}_max{max_replans}_handoff{handoff}"
        key += "_again" if key in res else ""
        res[key] = dict(ms=round(t, 4), kept=int(keep.numel()), same_as_first=same)
        print(key, res[key], flush=True)
        assert same
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
