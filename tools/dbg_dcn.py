import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, vision_amd
from helpers import gen
from oracle import oracle as O
g = gen(77); DEV = "cuda"; opt = torch.ops.tvmi.set_option
for dt in (torch.bfloat16, torch.float16):
    B, C, H, W, OC, groups, ogroups, stride, dil = 2, 64, 23, 31, 256, 1, 1, 1, 1
    x = torch.randn(B, C, H, W, generator=g).to(DEV, dt)
    w = (torch.randn(OC, C // groups, 3, 3, generator=g) * 0.1).to(DEV, dt)
    off = (torch.randn(B, 18, H, W, generator=g) * 3).to(DEV, dt)
    msk = torch.rand(B, 9, H, W, generator=g).to(DEV, dt)
    bias = torch.randn(OC, generator=g).to(DEV, dt)
    for m in (None, msk):
        opt("dcn.channels_last_gather", 0); want = vision_amd.deform_conv2d(x, off, w, bias, padding=1, mask=m)
        opt("dcn.channels_last_gather", 1); got = vision_amd.deform_conv2d(x, off, w, bias, padding=1, mask=m)
        d = (got.float() - want.float()).abs()
        idx = torch.nonzero(d > 0)
        O.load_reference()
        ref = torch.ops.torchvision.deform_conv2d(x.float().cpu(), w.float().cpu(), off.float().cpu(), (m if m is not None else torch.zeros(B, 1)).float().cpu(), bias.float().cpu(), 1, 1, 1, 1, 1, 1, 1, 1, m is not None)
        print(dt, m is not None, "mismatches", idx.shape[0], "max", float(d.max()), "planar err", float((want.float().cpu() - ref).abs().max()), "cl err", float((got.float().cpu() - ref).abs().max()))
        if idx.shape[0]:
            print(" first", idx[:6].tolist(), " distinct pixels", len(set((int(i[0]), int(i[2]), int(i[3])) for i in idx)), "distinct oc", len(set(int(i[1]) for i in idx)))
