"""RoIAlign forward timing under footprint variations (is the kernel memory-latency or issue bound?)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, vision_amd, bench
from vision_amd.poolers import _convert_to_roi_format
dev = torch.device("cuda:0")
def tm(fn, n=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
feats, boxes, scores = bench.make_inputs(dev, 1000)
scales = [1 / s for s in bench.STRIDES]
fl = [feats[str(i)] for i in range(4)]
def run(bx, tag, P=7):
    rois = _convert_to_roi_format(bx)
    t = tm(lambda: torch.ops.tvmi.multiscale_roi_align(fl, rois, scales, P, P, 2, False, 2, 5, 224.0, 4.0, 1e-6))
    print(f"{tag:40s} P={P}: {t:.4f} ms")
for P in (7, 14):
    run(boxes, "baseline", P)
    # same sizes, centres squeezed into 10 % of the image, all on image 0 -> L2-resident footprint
    sq = []
    for b in boxes:
        c = (b[:, :2] + b[:, 2:]) / 2; wh = b[:, 2:] - b[:, :2]
        c = c * 0.1 + 300
        sq.append(torch.cat([c - wh / 2, c + wh / 2], 1).clamp(min=0))
    allb = torch.cat(sq)
    z = torch.zeros(0, 4, device=dev)
    run([allb, z, z, z], "squeezed centres, image 0 (L2 resident)", P)
    one = boxes[0][:1].repeat(4000, 1)
    run([one, z, z, z], "one RoI x4000 (L1 resident)", P)
    # sorted by level+position (locality)
    key = []
    for b in boxes:
        key.append(b)
    run([b[torch.argsort((b[:, 1] // 64) * 100 + b[:, 0] // 64)] for b in boxes], "sorted by 64px cell", P)
# channels_last maps through the NHWC kernel
fl_nhwc = [f.contiguous(memory_format=torch.channels_last) for f in fl]
rois = _convert_to_roi_format(boxes)
a = torch.ops.tvmi.multiscale_roi_align(fl, rois, scales, 7, 7, 2, False, 2, 5, 224.0, 4.0, 1e-6)
b = torch.ops.tvmi.multiscale_roi_align(fl_nhwc, rois, scales, 7, 7, 2, False, 2, 5, 224.0, 4.0, 1e-6)
print("NHWC vs NCHW max abs diff:", (a - b).abs().max().item())
t = tm(lambda: torch.ops.tvmi.multiscale_roi_align(fl_nhwc, rois, scales, 7, 7, 2, False, 2, 5, 224.0, 4.0, 1e-6))
print(f"channels_last multiscale roi_align 7x7: {t:.4f} ms  ({566.35e6 / t / 1e6:.0f} GB/s algorithmic, {566.35e6 / t / 1e6 / 8000:.3f} of HBM peak)")
tc = tm(lambda: [f.contiguous() for f in fl_nhwc])
print(f"(reference route: NHWC->NCHW copies of the 4 maps alone {tc:.4f} ms)")
for dt in (torch.float16, torch.bfloat16):
    fh = [f.to(dt).contiguous(memory_format=torch.channels_last) for f in fl]
    rh = rois.to(dt)
    t = tm(lambda: torch.ops.tvmi.multiscale_roi_align(fh, rh, scales, 7, 7, 2, False, 2, 5, 224.0, 4.0, 1e-6))
    print(f"channels_last multiscale roi_align 7x7 {dt}: {t:.4f} ms  ({283.2e6 / t / 1e6:.0f} GB/s algorithmic)")
# locality experiment for the NHWC kernel: boxes sorted by (image, 64-px cell) before the call
def lvl_sorted(bx):
    out = []
    for b in bx:
        out.append(b[torch.argsort((b[:, 1] // 64) * 100 + b[:, 0] // 64)])
    return out
for name, bx in (("as given", boxes), ("sorted by 64px cell", lvl_sorted(boxes))):
    r = _convert_to_roi_format(bx)
    t = tm(lambda: torch.ops.tvmi.multiscale_roi_align(fl_nhwc, r, scales, 7, 7, 2, False, 2, 5, 224.0, 4.0, 1e-6))
    print(f"channels_last fp32, rois {name}: {t:.4f} ms")
# sort by level first, then position (RoIs of one level share maps)
from vision_amd.poolers import LevelMapper
def lvl_pos_sorted(bx):
    out = []
    lm = LevelMapper(2, 5)
    for b in bx:
        lv = lm([b])
        out.append(b[torch.argsort(lv * 10000 + (b[:, 1] // 64) * 100 + b[:, 0] // 64)])
    return out
r = _convert_to_roi_format(lvl_pos_sorted(boxes))
t = tm(lambda: torch.ops.tvmi.multiscale_roi_align(fl_nhwc, r, scales, 7, 7, 2, False, 2, 5, 224.0, 4.0, 1e-6))
print(f"channels_last fp32, rois sorted by (level, cell): {t:.4f} ms")
