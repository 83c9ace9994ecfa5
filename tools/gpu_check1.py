"""First GPU bring-up: parity of nms / roi_align vs the compiled reference CPU kernels + timing."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vision_amd
torch.ops.load_library(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle/_ref/libtv_ref_cpu.so"))
dev = "cuda"
print("device", torch.cuda.get_device_name(0), "hip", torch.version.hip, "cpus", os.cpu_count())

def boxes(n, W, H, lo, hi, g):
    xy = torch.rand(n, 2, generator=g) * torch.tensor([W - 64.0, H - 64.0])
    wh = lo + torch.rand(n, 2, generator=g) * (hi - lo)
    return torch.cat([xy, torch.minimum(xy + wh, torch.tensor([float(W), float(H)]))], 1)

def tm(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n

g = torch.Generator().manual_seed(0)
# ---- NMS
for n, thr in [(1000, 0.5), (5000, 0.7), (100000, 0.5)]:
    b = boxes(n, 1000, 1000, 1, 101, g) if n == 100000 else boxes(n, 1088, 800, 16, 300, g)
    s = torch.rand(n, generator=g)
    if n <= 5000:
        t0 = time.perf_counter(); kc = torch.ops.torchvision.nms(b, s, thr); tc = time.perf_counter() - t0
    else:
        kc = None; tc = float("nan")
    kg = torch.ops.torchvision.nms(b.to(dev), s.to(dev), thr)
    ok = None if kc is None else bool(torch.equal(kc, kg.cpu()))
    t = tm(lambda: torch.ops.torchvision.nms(b.to(dev), s.to(dev), thr), n=10)
    print(f"nms n={n} kept={kg.numel()} exact={ok} gpu_ms={t:.3f} cpu_ms={tc*1e3:.2f}")
# ---- RoIAlign
x = torch.randn(1, 256, 200, 272, generator=g)
rb = boxes(1000, 1088, 800, 16, 300, g)
rois = torch.cat([torch.zeros(1000, 1), rb], 1)
for (ph, sr, al) in [(7, 2, False), (14, 2, False), (7, 0, True), (5, 3, True)]:
    t0 = time.perf_counter(); yc = torch.ops.torchvision.roi_align(x, rois, 0.25, ph, ph, sr, al); tc = time.perf_counter() - t0
    xg, rg = x.to(dev), rois.to(dev)
    yg = torch.ops.torchvision.roi_align(xg, rg, 0.25, ph, ph, sr, al)
    err = (yg.cpu() - yc).abs().max().item()
    t = tm(lambda: torch.ops.torchvision.roi_align(xg, rg, 0.25, ph, ph, sr, al))
    by = x.numel() * 4 + yc.numel() * 4
    print(f"roi_align {ph}x{ph} sr={sr} al={al} maxerr={err:.2e} gpu_ms={t:.4f} cpu_ms={tc*1e3:.1f} algGB/s={by/t/1e6:.0f}")
    gr = torch.randn(yc.shape, generator=g)
    gc = torch.ops.torchvision._roi_align_backward(gr, rois, 0.25, ph, ph, 1, 256, 200, 272, sr, al)
    gg = torch.ops.torchvision._roi_align_backward(gr.to(dev), rg, 0.25, ph, ph, 1, 256, 200, 272, sr, al)
    errb = (gg.cpu() - gc).abs().max().item()
    grd = gr.to(dev)
    tb = tm(lambda: torch.ops.torchvision._roi_align_backward(grd, rg, 0.25, ph, ph, 1, 256, 200, 272, sr, al))
    print(f"   bwd maxerr={errb:.2e} (scale {gc.abs().max().item():.1f}) gpu_ms={tb:.4f}")
for dt in (torch.bfloat16, torch.float16, torch.float64):
    xs = x[:, :32].to(dt); rs = rois[:200].to(dt)
    yc = torch.ops.torchvision.roi_align(xs.float() if dt != torch.float64 else xs, rs.float() if dt != torch.float64 else rs, 0.25, 7, 7, 2, False)
    yg = torch.ops.torchvision.roi_align(xs.to(dev), rs.to(dev), 0.25, 7, 7, 2, False)
    print(dt, "maxerr", (yg.cpu().to(yc.dtype) - yc).abs().max().item())
