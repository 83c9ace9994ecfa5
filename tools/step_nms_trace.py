"""Driver for a kernel trace of the detector step's NMS + payload: `python tools/step_nms_trace.py [fused|chain] [calls]`
(bench.py's config-2 boxes: 4 x 1000 proposals, segment = image)."""
import sys

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import vision_amd  # noqa: E402
from vision_amd import sharding  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "fused"
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device("cuda")
_, boxes, scores = bench.make_inputs(dev, 1000)
b, s = torch.cat(boxes), torch.cat(scores)
img = torch.arange(4, device=dev).repeat_interleave(1000)
torch.ops.tvmi.set_option("nms.step_fused", 1 if mode == "fused" else 0)
for i in range(calls):
    if mode == "fused":
        out = sharding.nms_pack_payload(b, s, img, 0.5, 4, img, 4, 100)
    else:
        k, n = vision_amd.boxes.batched_nms_padded(b, s, img, 0.5, 4)
        out = sharding.pack_kept_payload(b, s, img, k, n, 4, 100)
    if i % 8 == 7:
        torch.cuda.synchronize()
torch.cuda.synchronize()
print(mode, int(out[1] if mode == "fused" else n))
