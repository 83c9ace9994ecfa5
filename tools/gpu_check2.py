"""GPU bring-up #2: roi_pool / ps_roi_* / deform_conv2d / box_iou_rotated / resize vs reference CPU."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
import vision_amd
torch.ops.load_library(os.path.join(ROOT, "oracle/_ref/libtv_ref_cpu.so"))
dev = "cuda"
tv = torch.ops.torchvision
g = torch.Generator().manual_seed(0)

def tm(fn, n=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n

def cmp(name, a, b):
    a = a.cpu(); 
    if a.dtype.is_floating_point:
        e = (a.double() - b.double()).abs().max().item() if a.numel() else 0.0
        print(f"{name}: maxerr={e:.3e} (ref max {b.abs().max().item() if b.numel() else 0:.2f})")
    else:
        print(f"{name}: equal={bool(torch.equal(a, b))}")

x = torch.randn(2, 50 * 4, 24, 30, generator=g)
rois = torch.tensor([[0, 0, 0, 9, 9], [0, 0, 5, 4, 9], [0, 5, 5, 9, 9], [1, 0, 0, 9, 9], [1, 3.3, 2.7, 25.1, 19.9], [0, -3, -2, 40, 30]], dtype=torch.float32)
rois = torch.cat([rois, torch.cat([torch.randint(0, 2, (200, 1), generator=g).float(), torch.rand(200, 2, generator=g) * 20, 5 + torch.rand(200, 2, generator=g) * 25], 1)])
for scale in (1.0, 0.5):
    o, a = tv.roi_pool(x, rois, scale, 5, 5); og, ag = tv.roi_pool(x.to(dev), rois.to(dev), scale, 5, 5)
    cmp(f"roi_pool s={scale} out", og, o); cmp("  argmax", ag, a)
    gr = torch.randn(o.shape, generator=g)
    cmp("  bwd", tv._roi_pool_backward(gr.to(dev), rois.to(dev), ag, scale, 5, 5, 2, 200, 24, 30), tv._roi_pool_backward(gr, rois, a, scale, 5, 5, 2, 200, 24, 30))
    for sr in (2, 0):
        o, m = tv.ps_roi_align(x, rois, scale, 5, 5, sr); og, mg = tv.ps_roi_align(x.to(dev), rois.to(dev), scale, 5, 5, sr)
        cmp(f"ps_roi_align s={scale} sr={sr} out", og, o); cmp("  mapping", mg, m)
        gr = torch.randn(o.shape, generator=g)
        cmp("  bwd", tv._ps_roi_align_backward(gr.to(dev), rois.to(dev), mg, scale, 5, 5, sr, 2, 200, 24, 30), tv._ps_roi_align_backward(gr, rois, m, scale, 5, 5, sr, 2, 200, 24, 30))
    o, m = tv.ps_roi_pool(x, rois, scale, 5, 5); og, mg = tv.ps_roi_pool(x.to(dev), rois.to(dev), scale, 5, 5)
    cmp(f"ps_roi_pool s={scale} out", og, o); cmp("  mapping", mg, m)
    gr = torch.randn(o.shape, generator=g)
    cmp("  bwd", tv._ps_roi_pool_backward(gr.to(dev), rois.to(dev), mg, scale, 5, 5, 2, 200, 24, 30), tv._ps_roi_pool_backward(gr, rois, m, scale, 5, 5, 2, 200, 24, 30))

# deform conv: reference test config (asymmetric), mfma-sized config, depthwise
def dcn_case(B, C, OC, H, W, kh, kw, groups, og, stride, pad, dil, use_mask, time_it=False, bwd=True):
    oh = (H + 2 * pad[0] - (dil[0] * (kh - 1) + 1)) // stride[0] + 1
    ow = (W + 2 * pad[1] - (dil[1] * (kw - 1) + 1)) // stride[1] + 1
    x = torch.randn(B, C, H, W, generator=g); w = torch.randn(OC, C // groups, kh, kw, generator=g) * 0.05
    off = torch.randn(B, 2 * og * kh * kw, oh, ow, generator=g); m = torch.rand(B, og * kh * kw, oh, ow, generator=g) if use_mask else torch.zeros(B, 1)
    b = torch.randn(OC, generator=g)
    args = (stride[0], stride[1], pad[0], pad[1], dil[0], dil[1], groups, og, use_mask)
    t0 = time.perf_counter(); yc = tv.deform_conv2d(x, w, off, m, b, *args); tc = time.perf_counter() - t0
    xg, wg, og_, mg, bg = [t.to(dev) for t in (x, w, off, m, b)]
    yg = tv.deform_conv2d(xg, wg, og_, mg, bg, *args)
    cmp(f"deform_conv2d B{B} C{C} OC{OC} {H}x{W} k{kh}x{kw} g{groups} og{og} mask={use_mask}", yg, yc)
    if time_it:
        t = tm(lambda: tv.deform_conv2d(xg, wg, og_, mg, bg, *args))
        fl = 2.0 * B * OC * (C // groups) * kh * kw * oh * ow
        print(f"   gpu_ms={t:.3f} cpu_ms={tc*1e3:.0f} TFLOP/s={fl/t/1e9:.2f}")
    if bwd:
        gr = torch.randn(yc.shape, generator=g)
        rc = tv._deform_conv2d_backward(gr, x, w, off, m, b, *args)
        rg = tv._deform_conv2d_backward(gr.to(dev), xg, wg, og_, mg, bg, *args)
        for nm, a_, b_ in zip(("gin", "gw", "goff", "gmask", "gbias"), rg, rc): cmp("   bwd " + nm, a_, b_)

dcn_case(33, 6, 2, 5, 4, 3, 2, 2, 3, (2, 1), (1, 0), (2, 1), True)
dcn_case(3, 6, 2, 5, 4, 3, 2, 2, 3, (2, 1), (1, 0), (2, 1), False)
dcn_case(2, 64, 96, 20, 24, 3, 3, 1, 2, (1, 1), (1, 1), (1, 1), True)
dcn_case(2, 48, 40, 20, 24, 3, 3, 2, 4, (1, 1), (1, 1), (1, 1), False)
dcn_case(2, 256, 256, 100, 136, 3, 3, 1, 1, (1, 1), (1, 1), (1, 1), False, time_it=True, bwd=False)
dcn_case(2, 256, 256, 100, 136, 3, 3, 256, 1, (1, 1), (1, 1), (1, 1), False, time_it=True, bwd=False)

# rotated iou
def rboxes(n):
    return torch.cat([torch.rand(n, 2, generator=g) * 100, 5 + torch.rand(n, 2, generator=g) * 40, torch.rand(n, 1, generator=g) * 360 - 180], 1)
b1, b2 = rboxes(300), rboxes(257)
b1[:5] = b2[:5]
cmp("box_iou_rotated f32", tv.box_iou_rotated(b1.to(dev), b2.to(dev)), tv.box_iou_rotated(b1, b2))
cmp("box_iou_rotated f64", tv.box_iou_rotated(b1.double().to(dev), b2.double().to(dev)), tv.box_iou_rotated(b1.double(), b2.double()))
ka = torch.tensor([[0, 0, 10, 10, 45.], [0, 0, 10, 10, 0], [5, 5, 10, 10, 0]]); kb = torch.tensor([[0, 0, 10, 10, 135.], [0, 0, 10, 10, 0], [20, 20, 2, 2, 0]])
print(tv.box_iou_rotated(ka.to(dev), kb.to(dev)).cpu(), tv.box_iou_rotated(ka, kb))

# resize
img = torch.rand(2, 3, 120, 160, generator=g)
for mode in ("nearest", "nearest-exact", "bilinear", "bicubic"):
    for size in ((77, 99), (240, 333), (120, 160), (60, 80)):
        for aa in ((False, True) if mode in ("bilinear", "bicubic") else (False,)):
            for ac in ((False, True) if mode in ("bilinear", "bicubic") else (None,)):
                ref = F.interpolate(img, size=size, mode=mode, align_corners=ac, antialias=aa)
                out = vision_amd.interpolate(img.to(dev), size=size, mode=mode, align_corners=ac, antialias=aa)
                e = (out.cpu() - ref).abs().max().item()
                if e > 1e-5: print(f"resize {mode} {size} aa={aa} ac={ac} maxerr={e:.2e}  <<<<")
    ref = F.interpolate(img, scale_factor=1.7, mode=mode); out = vision_amd.interpolate(img.to(dev), scale_factor=1.7, mode=mode)
    ref2 = F.interpolate(img, scale_factor=0.6, mode=mode, recompute_scale_factor=True); out2 = vision_amd.interpolate(img.to(dev), scale_factor=0.6, mode=mode, recompute_scale_factor=True)
    print(f"resize {mode} scale_factor: {(out.cpu()-ref).abs().max().item():.2e} {(out2.cpu()-ref2).abs().max().item():.2e}")
big = torch.rand(8, 3, 1080, 1920, generator=g).to(dev)
for mode, aa in (("bilinear", False), ("bilinear", True), ("bicubic", False), ("bicubic", True)):
    t = tm(lambda: vision_amd.interpolate(big, size=(800, 1422), mode=mode, antialias=aa))
    t2 = tm(lambda: F.interpolate(big, size=(800, 1422), mode=mode, antialias=aa))
    by = big.numel() * 4 + 8 * 3 * 800 * 1422 * 4
    print(f"resize 8x3x1080x1920->800x1422 {mode} aa={aa}: ours {t:.3f} ms ({by/t/1e6:.0f} GB/s)  aten {t2:.3f} ms")
print("done")
