import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, vision_amd
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
xy = torch.rand(n, 2, generator=g) * 1000; wh = 8 + torch.rand(n, 2, generator=g) * 120
b = torch.cat([xy, xy + wh], 1).to(dev); s = torch.rand(n, generator=g).to(dev)
for _ in range(5):
    k = torch.ops.torchvision.nms(b, s, 0.5)
torch.cuda.synchronize()
print("kept", len(k))
