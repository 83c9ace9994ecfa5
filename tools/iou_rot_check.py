"""Rotated IoU against the C restatement on fuzz-shaped inputs (clustered / identical / axis-aligned boxes): prints the worst
pair of every failing case.  python tools/iou_rot_check.py [seed] [cases]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, vision_amd
from oracle import oracle as O
dev = torch.device("cuda:0"); tv = torch.ops.torchvision
g = torch.Generator().manual_seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
def ri(a, b): return int(torch.randint(a, b + 1, (1,), generator=g))
worst = 0.0; bad = 0
for case in range(int(sys.argv[2]) if len(sys.argv) > 2 else 300):
    n1, n2, spread = ri(1, 200), ri(1, 200), [30.0, 300.0, 3000.0][ri(0, 2)]
    r1 = torch.cat([torch.rand(n1, 2, generator=g) * spread, 1 + torch.rand(n1, 2, generator=g) * 80, torch.rand(n1, 1, generator=g) * 720 - 360], 1)
    r2 = torch.cat([torch.rand(n2, 2, generator=g) * spread, 1 + torch.rand(n2, 2, generator=g) * 80, torch.rand(n2, 1, generator=g) * 720 - 360], 1)
    ident, axis = ri(0, 3) == 0, ri(0, 3) == 0
    if ident: r2[: min(n1, n2)] = r1[: min(n1, n2)]
    if axis: r1[:, 4] = r1[:, 4].round() * 90
    iou = tv.box_iou_rotated(r1.to(dev), r2.to(dev)).cpu().numpy()
    ref = O.box_iou_rotated(r1.numpy(), r2.numpy())
    err = np.abs(iou - ref)
    worst = max(worst, float(err.max()))
    if err.max() >= 1e-5:
        bad += 1
        i, j = np.unravel_index(err.argmax(), err.shape)
        # what the reference's float32 arithmetic gives when an input moves by a float ulp or two
        outs = set()
        for which in (0, 1):
            for col in (0, 1, 2, 3, 4):
                for d in (-2, -1, 1, 2):
                    a, b2_ = r1[i:i + 1].clone().numpy(), r2[j:j + 1].clone().numpy()
                    t = a if which == 0 else b2_
                    v = t[0, col]
                    for _ in range(abs(d)):
                        v = np.nextafter(v, np.float32(np.inf if d > 0 else -np.inf), dtype=np.float32)
                    t[0, col] = v
                    outs.add(round(float(O.box_iou_rotated(a, b2_)[0, 0]), 5))
        exact = float(O.box_iou_rotated(r1[i:i + 1].double().numpy(), r2[j:j + 1].double().numpy())[0, 0])
        print(f"   float64: {exact:.7f}; float32 reference under +-1..2 ulp input perturbations: {sorted(outs)}")
        print(f"case {case}: n1={n1} n2={n2} spread={spread} ident={ident} axis={axis} max err {err.max():.3e} at ({i},{j}): gpu {iou[i, j]:.7f} ref {ref[i, j]:.7f} "
              f"count>=1e-5: {(err >= 1e-5).sum()}\n   b1={r1[i].tolist()}\n   b2={r2[j].tolist()}", flush=True)
print(f"worst {worst:.3e}, {bad} failing cases")
