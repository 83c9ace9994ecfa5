import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, vision_amd
from helpers import golden
lib = vision_amd._loader.kernels(); tv = torch.ops.torchvision
g = golden("roi_ops"); x = torch.as_tensor(g["x"]).cuda(); rois = torch.as_tensor(g["rois"]).cuda()
for variant in (0, 1):
    lib.tvmi_debug_set(0, variant)
    for scale in (1.0, 0.5):
        for sr in (-1, 2):
            for al in (False, True):
                y = tv.roi_align(x, rois, scale, 5, 5, sr, al).cpu().numpy()
                ref = g[f"roi_align_s{scale}_sr{sr}_a{int(al)}"]
                err = np.abs(y - ref).reshape(len(rois), -1).max(1)
                bad = np.nonzero(err > 1e-5)[0]
                if len(bad): print(f"variant {variant} scale {scale} sr {sr} al {al}: bad rois {bad.tolist()} {g['rois'][bad].tolist()} err {err[bad]}")
print("done")
