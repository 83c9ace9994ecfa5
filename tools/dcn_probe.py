"""deform_conv2d config 4 timing (g=1 and depthwise), 4-wave vs 8-wave tiles via TVMI_DCN_WAVES."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, vision_amd
dev = torch.device("cuda:0")
def tm(fn, n=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
g = torch.Generator().manual_seed(0)
x = torch.randn(2, 256, 100, 136, generator=g).to(dev)
off = torch.randn(2, 18, 100, 136, generator=g).to(dev)
msk = torch.rand(2, 9, 100, 136, generator=g).to(dev)
for groups, oc in ((1, 256), (1, 128), (1, 64), (256, 256)):
    w = (torch.randn(oc, 256 // groups, 3, 3, generator=g) * 0.01).to(dev); b = torch.zeros(oc, device=dev)
    f = lambda: torch.ops.torchvision.deform_conv2d(x, w, off, msk, b, 1, 1, 1, 1, 1, 1, groups, 1, True)
    t = tm(f)
    fl = 2 * 2 * oc * (256 // groups) * 9 * 100 * 136
    print(f"deform_conv2d 2x256x100x136 -> {oc} ch, groups={groups}: {t:.4f} ms  {fl / t / 1e9:.1f} TFLOP/s  (TVMI_DCN_WAVES={os.environ.get('TVMI_DCN_WAVES', 'auto')})")
# a problem that fills the chip many times over
x = torch.randn(4, 256, 200, 272, generator=g).to(dev); off = torch.randn(4, 18, 200, 272, generator=g).to(dev)
w = (torch.randn(256, 256, 3, 3, generator=g) * 0.01).to(dev); b = torch.zeros(256, device=dev)
t = tm(lambda: torch.ops.torchvision.deform_conv2d(x, w, off, torch.zeros(4, 1, device=dev), b, 1, 1, 1, 1, 1, 1, 1, 1, False), n=5)
fl = 2 * 4 * 256 * 256 * 9 * 200 * 272
print(f"deform_conv2d 4x256x200x272 -> 256 ch: {t:.4f} ms  {fl / t / 1e9:.1f} TFLOP/s  (TVMI_DCN_WAVES={os.environ.get('TVMI_DCN_WAVES', 'auto')})")
