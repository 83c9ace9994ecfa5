"""batched_nms 100k x 80 (BASELINE config 3): a few calls for a rocprofv3 --kernel-trace --stats pass."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, vision_amd
dev = "cuda"; n = 100_000
g = torch.Generator().manual_seed(7)
xy = torch.rand(n, 2, generator=g) * 936; wh = 1 + torch.rand(n, 2, generator=g) * 100
b = torch.cat([xy, torch.minimum(xy + wh, torch.tensor([1000.0, 1000.0]))], 1).to(dev)
s = torch.rand(n, generator=g).to(dev); idx = torch.randint(0, 80, (n,), generator=g).to(dev)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    vision_amd.batched_nms(b, s, idx, 0.5)
torch.cuda.synchronize()
