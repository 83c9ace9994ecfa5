"""Randomised parity sweep against the oracle (NMS paths incl. the chunked large path and degenerate boxes, RoIAlign NCHW /
channels_last / tile-owner backward 7x7 + 14x14 / the register-staged kernels of generic pooled shapes, RoIPool column kernel,
PS ops, resize forward + channels_last + backward against torch CPU, rotated IoU); 60 s of cases on the box.  python tools/fuzz_gpu.py [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, vision_amd
from oracle import oracle as O
dev = torch.device("cuda:0"); tv = torch.ops.torchvision
g = torch.Generator().manual_seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
def ri(a, b): return int(torch.randint(a, b + 1, (1,), generator=g))
t0 = time.time(); cases = 0; unstable_pairs = 0
while time.time() - t0 < 60:
    # ---- NMS / batched NMS
    n = ri(1, 6000) if ri(0, 7) else ri(6000, 25000); canvas = float(ri(20, 800)); S = ri(1, 40)
    xy = torch.rand(n, 2, generator=g) * canvas; wh = torch.rand(n, 2, generator=g) * ri(2, 120)
    b = torch.cat([xy, xy + wh], 1)
    if ri(0, 5) == 0:                                     # degenerate boxes: zero area, inverted, tiny / huge scale
        k0 = ri(3, 50); b[::k0, 2:] = b[::k0, :2]
        k1 = ri(3, 50); b[1::k1, 2] = b[1::k1, 0] - 1.0
        if ri(0, 1): b[2::ri(5, 60)] *= 2.0 ** -38
        if ri(0, 1): b[3::ri(5, 60)] *= 2.0 ** 33
    if ri(0, 4) == 0: b -= canvas * 0.6                  # coordinates below -1: the coordinate trick of batched_nms suppresses across categories
    s = torch.rand(n, generator=g)
    if ri(0, 2) == 0: s = (s * ri(2, 50)).floor() / 16
    idx = torch.randint(0, S, (n,), generator=g)
    thr = [0.3, 0.5, 0.7, float(torch.rand(1, generator=g))][ri(0, 3)]
    if n > 4200:   # the large path re-plans on its survivors: fuzz the threshold, the first-phase share and the depth
        torch.ops.tvmi.set_option("nms.replan_min_boxes", [0, 4097, ri(4097, n), 24576][ri(0, 3)])
        torch.ops.tvmi.set_option("nms.replan_divisor", ri(2, 40)); torch.ops.tvmi.set_option("nms.replan_max", ri(1, 5))
        torch.ops.tvmi.set_option("nms.device_handoff", ri(0, 1))
    want = O.nms(b.numpy(), s.numpy(), thr)
    got = tv.nms(b.to(dev), s.to(dev), thr).cpu().numpy()
    assert np.array_equal(got, want), ("nms", n, canvas, thr)
    wants = O.nms(b.numpy(), s.numpy(), thr, idx.numpy())
    for hint in (-1, S):
        gots = torch.ops.tvmi.nms_segmented(b.to(dev), s.to(dev), idx.to(dev), thr, hint).cpu().numpy()
        assert np.array_equal(gots, wants), ("segmented", n, S, hint, thr)
        gotb = vision_amd.batched_nms(b.to(dev), s.to(dev), idx.to(dev), thr, num_segments=hint).cpu().numpy()   # the reference's switch of arithmetic
        assert np.array_equal(gotb, O.batched_nms(b, s, idx, thr)), ("batched", n, S, hint, thr)
    # ---- round 6: the one-launch step kernel with its payload (n <= 4096, <= 64 segments of <= 1024, <= 16 images) against the oracle
    # and against the launch chain + pack_detections_payload
    if n <= 4096 and int(torch.bincount(idx, minlength=S).max()) <= 1024:
        Bi = ri(1, min(16, S)); img = (idx % Bi).to(dev); D = ri(1, 120); lab = torch.randint(0, 91, (n,), generator=g).to(dev)
        db, ds, di = b.to(dev), s.to(dev), idx.to(dev)
        k1, n1, p1 = torch.ops.tvmi.nms_step(db, ds, di, thr, S, img, lab, Bi, D)
        assert int(n1) == len(wants) and np.array_equal(k1[: int(n1)].cpu().numpy(), wants), ("nms_step", n, S, thr)
        torch.ops.tvmi.set_option("nms.step_fused", 0)
        k2, n2 = torch.ops.tvmi.nms_segmented_padded(db, ds, di, thr, S)
        torch.ops.tvmi.set_option("nms.step_fused", 1)
        assert int(n2) == int(n1) and torch.equal(k2[: int(n2)], k1[: int(n1)]), ("nms chain", n, S, thr)
        assert torch.equal(p1, torch.ops.tvmi.pack_detections_payload(db, ds, lab, img, k2, n2, Bi, D)), ("nms_step payload", n, S, Bi, D)
    # ---- RoIAlign forward NCHW vs channels_last vs oracle, backward vs oracle
    N, C, H, W = ri(1, 3), [ri(1, 70), ri(1, 70), 256, 512][ri(0, 3)], ri(2, 60), ri(4, 70)
    # launch routes of the LDS-DMA forward (C = 256 / 512: channel chunks pinned to XCDs, launch order from the pre-pass)
    torch.ops.tvmi.set_option("roi_align.pin_chunks", ri(0, 1)); torch.ops.tvmi.set_option("roi_align.order", ri(0, 1))
    torch.ops.tvmi.set_option("roi_align.order_bands", [1, 16, 64][ri(0, 2)]); torch.ops.tvmi.set_option("roi_align.inline_mop", ri(0, 1))
    x = torch.rand(N, C, H, W, generator=g)
    k = ri(1, 60); scale = [1.0, 0.5, 0.25][ri(0, 2)]
    bx = torch.rand(k, 2, generator=g) * torch.tensor([W / scale, H / scale]) * 1.1 - 0.05 * W / scale
    bw = torch.rand(k, 2, generator=g) * torch.tensor([W / scale, H / scale]) * [0.1, 0.5, 1.2][ri(0, 2)]
    rois = torch.cat([torch.randint(0, N, (k, 1), generator=g).float(), bx, bx + bw], 1)
    aligned = bool(ri(0, 1))
    ref = O.roi_align(x.numpy(), rois.numpy(), scale, 7, 7, 2, aligned)
    y = tv.roi_align(x.to(dev), rois.to(dev), scale, 7, 7, 2, aligned)
    assert np.abs(y.cpu().numpy() - ref).max() < 1e-4, ("roi nchw", N, C, H, W)
    if C > 1:
        ycl = tv.roi_align(x.to(dev).contiguous(memory_format=torch.channels_last), rois.to(dev), scale, 7, 7, 2, aligned)
        assert np.abs(ycl.cpu().numpy() - ref).max() < 1e-4, ("roi nhwc", N, C, H, W)
    gr = torch.randn(ref.shape, generator=g)
    gi = tv._roi_align_backward(gr.to(dev), rois.to(dev), scale, 7, 7, N, C, H, W, 2, aligned)
    refb = O.roi_align_backward(gr.numpy(), rois.numpy(), scale, 7, 7, N, C, H, W, 2, aligned)
    assert np.abs(gi.cpu().numpy() - refb).max() < 1e-4 * max(1.0, float(np.abs(refb).max())), ("roi bwd", N, C, H, W)
    if ri(0, 3) == 0:                                     # 14x14 backward (owner kernel, 14x14 instantiation)
        gr = torch.randn(k, C, 14, 14, generator=g)
        gi = tv._roi_align_backward(gr.to(dev), rois.to(dev), scale, 14, 14, N, C, H, W, 2, aligned)
        refb = O.roi_align_backward(gr.numpy(), rois.numpy(), scale, 14, 14, N, C, H, W, 2, aligned)
        assert np.abs(gi.cpu().numpy() - refb).max() < 1e-4 * max(1.0, float(np.abs(refb).max())), ("roi bwd14", N, C, H, W)
    # ---- round 6: multi-scale RoIAlign from box lists with the order pre-pass FOLDED INTO THE LAUNCH (one workgroup sorts, the first
    # round of units runs in input order, later units wait for their order entries) against the two-launch form: same bits, same rows
    if ri(0, 2) == 0:
        nim = ri(1, 6); L = ri(2, 4); Hm, Wm = ri(24, 80), ri(32, 120)
        total = [ri(1, 900), ri(900, 1100), ri(1100, 4096), ri(4090, 4200)][ri(0, 3)]
        cuts = sorted(ri(0, total) for _ in range(nim - 1)); cnts = [b_ - a_ for a_, b_ in zip([0] + cuts, cuts + [total])]
        fm = [torch.randn(nim, 256, max(Hm >> l, 2), max(Wm >> l, 4), generator=g).to(dev) for l in range(L)]
        bl = []
        for m in cnts:
            o = torch.rand(m, 2, generator=g) * torch.tensor([Wm * 4.0, Hm * 4.0]) - 8.0
            e = torch.rand(m, 2, generator=g) * [16.0, 120.0, 500.0][ri(0, 2)]
            bb = torch.cat([o, o + e], 1)
            if m and ri(0, 3) == 0: bb[::3, 2:] = bb[::3, :2]                      # zero-area boxes
            if m and ri(0, 4) == 0: bb[ri(0, m - 1)] = float("nan")
            if m and ri(0, 4) == 0: bb[ri(0, m - 1), 2:] = bb[ri(0, m - 1), :2] - 3.0
            bl.append(bb.to(dev))
        sc = [0.25 / (1 << l) for l in range(L)]; tail = (7, 7, 2, bool(ri(0, 1)), 2, 1 + L, 224.0, 4.0, 1e-6)
        torch.ops.tvmi.set_option("roi_align.pin_chunks", 1); torch.ops.tvmi.set_option("roi_align.order", 1); torch.ops.tvmi.set_option("roi_align.inline_mop", 1)
        torch.ops.tvmi.set_option("roi_align.fold_first_round_pct", [0, 50, 100, 200][ri(0, 3)])
        torch.ops.tvmi.set_option("roi_align.fold_order", 1); fo, fr = torch.ops.tvmi.multiscale_roi_align_boxes(fm, bl, sc, *tail)
        torch.ops.tvmi.set_option("roi_align.fold_order", 0); uo, ur = torch.ops.tvmi.multiscale_roi_align_boxes(fm, bl, sc, *tail)
        torch.ops.tvmi.set_option("roi_align.fold_order", 1); torch.ops.tvmi.set_option("roi_align.fold_first_round_pct", 100)
        if not torch.equal(fr.view(torch.int32), ur.view(torch.int32)):
            want = torch.cat([torch.cat([torch.full((b_.shape[0], 1), float(i_), device=dev), b_], 1) for i_, b_ in enumerate(bl)])
            rb = (fr.view(torch.int32) != ur.view(torch.int32)).any(1).nonzero()[:, 0]
            print("fold rows differ:", rb.numel(), rb[:8].tolist(), "fold-vs-want", int((fr.view(torch.int32) != want.view(torch.int32)).any(1).sum()),
                  "unfold-vs-want", int((ur.view(torch.int32) != want.view(torch.int32)).any(1).sum()), "fold", fr[int(rb[0])].tolist(), "unfold", ur[int(rb[0])].tolist(),
                  "want", want[int(rb[0])].tolist(), flush=True)
        assert torch.equal(fr.view(torch.int32), ur.view(torch.int32)), ("fold rows", cnts)
        assert torch.equal(fo.view(torch.int32), uo.view(torch.int32)), ("fold out", cnts, L, Hm, Wm)
        cases += 1
    # ---- RoIPool 7x7 (column kernel): value and argmax bit-exact, ties from rounded values
    C2 = [ri(1, 40), ri(40, 130), ri(250, 300)][ri(0, 2)]
    xp = (torch.randn(N, C2, H, W, generator=g) * 2).round()
    if ri(0, 1): rois_p = rois[torch.argsort(rois[:, 0], stable=True)]
    else: rois_p = rois
    dt = [torch.float32, torch.float16, torch.bfloat16][ri(0, 2)]
    yp, ap = tv.roi_pool(xp.to(dt).to(dev), rois_p.to(dt).to(dev), scale, 7, 7)
    ryp, rap = O.roi_pool(xp.to(dt).float().numpy(), rois_p.to(dt).float().numpy(), scale, 7, 7)
    assert np.array_equal(ap.cpu().numpy(), rap) and np.array_equal(yp.float().cpu().numpy(), ryp), ("roi_pool", N, C2, H, W, dt)
    if dt == torch.float32:                               # RoIPool backward, plane-owner regime (or atomics for big planes)
        gp = torch.randn(k, C2, 7, 7, generator=g)
        gip = tv._roi_pool_backward(gp.to(dev), rois_p.to(dev), ap, scale, 7, 7, N, C2, H, W)
        rgp = O.roi_pool_backward(gp.numpy(), rois_p.numpy(), rap, N, C2, H, W)
        assert np.abs(gip.cpu().numpy() - rgp).max() < 1e-4 * max(1.0, float(np.abs(rgp).max())), ("roi_pool bwd", N, C2, H, W)
    # ---- PSRoIAlign / PSRoIPool forward + plane-owner backward
    P = ri(1, 4); Co = ri(1, 4); Cps = Co * P * P; srp = ri(0, 3)
    xs = torch.randn(N, Cps, H, W, generator=g)
    ya, ma = tv.ps_roi_align(xs.to(dev), rois.to(dev), scale, P, P, srp)
    rya, rma = O.ps_roi_align(xs.numpy(), rois.numpy(), scale, P, P, srp)
    assert np.abs(ya.cpu().numpy() - rya).max() < 1e-4 and np.array_equal(ma.cpu().numpy(), rma), ("ps_align", N, Cps, H, W, P, srp)
    gs = torch.randn(k, Co, P, P, generator=g)
    ga = tv._ps_roi_align_backward(gs.to(dev), rois.to(dev), ma, scale, P, P, srp, N, Cps, H, W)
    rga = O.ps_roi_align_backward(gs.numpy(), rois.numpy(), rma, scale, P, P, srp, N, Cps, H, W)
    assert np.abs(ga.cpu().numpy() - rga).max() < 1e-4 * max(1.0, float(np.abs(rga).max())), ("ps_align bwd", N, Cps, H, W, P, srp)
    yq, mq = tv.ps_roi_pool(xs.to(dev), rois.to(dev), scale, P, P)
    ryq, rmq = O.ps_roi_pool(xs.numpy(), rois.numpy(), scale, P, P)
    assert np.abs(yq.cpu().numpy() - ryq).max() < 1e-4 and np.array_equal(mq.cpu().numpy(), rmq), ("ps_pool", N, Cps, H, W, P)
    gq = tv._ps_roi_pool_backward(gs.to(dev), rois.to(dev), mq, scale, P, P, N, Cps, H, W)
    rgq = O.ps_roi_pool_backward(gs.numpy(), rois.numpy(), rmq, scale, P, P, N, Cps, H, W)
    assert np.abs(gq.cpu().numpy() - rgq).max() < 1e-4 * max(1.0, float(np.abs(rgq).max())), ("ps_pool bwd", N, Cps, H, W, P)
    # ---- register-staged RoIAlign forward (generic pooled shapes, adaptive sampling: the wave kernels), fp32 and 16-bit
    php, pwp, srr = ri(1, 9), ri(1, 9), [-1, 0, 1, 2, 3][ri(0, 4)]
    refw = O.roi_align(x.numpy(), rois.numpy(), scale, php, pwp, srr, aligned)
    yw = tv.roi_align(x.to(dev), rois.to(dev), scale, php, pwp, srr, aligned)
    assert np.abs(yw.cpu().numpy() - refw).max() < 1e-4, ("roi wave", N, C, H, W, php, pwp, srr)
    if ri(0, 2) == 0:
        d16 = [torch.float16, torch.bfloat16][ri(0, 1)]
        ref16 = O.roi_align(x.to(d16).float().numpy(), rois.to(d16).float().numpy(), scale, php, pwp, srr, aligned)
        y16 = tv.roi_align(x.to(d16).to(dev), rois.to(d16).to(dev), scale, php, pwp, srr, aligned)
        assert np.abs(y16.float().cpu().numpy() - ref16).max() < (4e-3 if d16 == torch.float16 else 2e-2), ("roi wave 16", d16, php, pwp, srr)
    # ---- resize: forward (NCHW + channels_last) and backward of the six modes against torch CPU
    import torch.nn.functional as F
    mode, aa = [("nearest", False), ("nearest-exact", False), ("bilinear", False), ("bicubic", False), ("bilinear", True), ("bicubic", True)][ri(0, 5)]
    Nr, Cr, ih, iw = ri(1, 2), ri(1, 9), ri(1, 70), ri(4, 300)
    oh, ow = ri(1, 140), ri(4, 400)
    kw = {} if mode.startswith("nearest") else dict(align_corners=bool(ri(0, 1)) and not aa, antialias=aa)
    xr = torch.rand(Nr, Cr, ih, iw, generator=g, requires_grad=True)
    yr = F.interpolate(xr, size=(oh, ow), mode=mode, **kw)
    wr = torch.randn(yr.shape, generator=g)
    (yr * wr).sum().backward()
    xd = xr.detach().to(dev).requires_grad_(True)
    yd = vision_amd.interpolate(xd, size=(oh, ow), mode=mode, **kw)
    assert float((yd.detach().cpu() - yr.detach()).abs().max()) < 1e-4, ("resize", mode, aa, kw, (ih, iw), (oh, ow))
    (yd * wr.to(dev)).sum().backward()
    gs_ = max(float(xr.grad.abs().max()), 1e-6)
    assert float((xd.grad.cpu() - xr.grad).abs().max()) / gs_ < 2e-5, ("resize bwd", mode, aa, kw, (ih, iw), (oh, ow))
    if Cr > 1:
        ycl = vision_amd.interpolate(xr.detach().to(dev).contiguous(memory_format=torch.channels_last), size=(oh, ow), mode=mode, **kw)
        assert float((ycl.cpu() - yr.detach()).abs().max()) < 1e-4, ("resize nhwc", mode, aa, kw, (ih, iw), (oh, ow))
    # ---- rotated IoU: random sizes (ragged 64 x 64 tiles), clustered or spread boxes
    n1, n2, spread = ri(1, 200), ri(1, 200), [30.0, 300.0, 3000.0][ri(0, 2)]
    r1 = torch.cat([torch.rand(n1, 2, generator=g) * spread, 1 + torch.rand(n1, 2, generator=g) * 80, torch.rand(n1, 1, generator=g) * 720 - 360], 1)
    r2 = torch.cat([torch.rand(n2, 2, generator=g) * spread, 1 + torch.rand(n2, 2, generator=g) * 80, torch.rand(n2, 1, generator=g) * 720 - 360], 1)
    if ri(0, 3) == 0: r2[: min(n1, n2)] = r1[: min(n1, n2)]          # identical boxes
    if ri(0, 3) == 0: r1[:, 4] = r1[:, 4].round() * 90                 # axis-aligned
    iou = tv.box_iou_rotated(r1.to(dev), r2.to(dev)).cpu().numpy()
    err = np.abs(iou - O.box_iou_rotated(r1.numpy(), r2.numpy()))
    if err.max() >= 1e-5:
        # The reference's float32 arithmetic is unstable on a few pairs per million of clustered boxes: moving ONE input by one or
        # two float ulps makes ITS OWN result jump between two values (0.0897 <-> 0.1970, 0.0636 <-> 0.0920, ...: one of them is
        # the float64 value, tools/iou_rot_check.py), and the device's cos / sin differ from glibc's in the last place.  On
        # such a pair the kernel must give one of the values the reference gives in that neighbourhood.
        off = np.argwhere(err >= 1e-5)
        assert len(off) <= 3, ("rotated", n1, n2, spread, len(off))
        for i_, j_ in off:
            outs = [float(O.box_iou_rotated(r1[i_:i_ + 1].double().numpy(), r2[j_:j_ + 1].double().numpy())[0, 0])]
            for which in (0, 1):
                for col in range(5):
                    for d in (-2, -1, 1, 2):
                        a_, b_ = r1[i_:i_ + 1].clone().numpy(), r2[j_:j_ + 1].clone().numpy()
                        t_ = a_ if which == 0 else b_
                        v_ = t_[0, col]
                        for _ in range(abs(d)):
                            v_ = np.nextafter(v_, np.float32(np.inf if d > 0 else -np.inf), dtype=np.float32)
                        t_[0, col] = v_
                        outs.append(float(O.box_iou_rotated(a_, b_)[0, 0]))
            assert min(abs(float(iou[i_, j_]) - o) for o in outs) < 2e-5, ("rotated", n1, n2, spread, float(iou[i_, j_]), sorted(set(round(o, 5) for o in outs)))
        unstable_pairs += len(off)
    cases += 1
print(f"fuzz ok: {cases} random cases (each: nms, 2x batched nms, the folded multi-scale RoIAlign launch against the two-launch form in a third of the cases, roi_align NCHW + channels_last + backward + generic-shape wave kernels, roi_pool fwd + bwd, "
      f"ps_roi_align / ps_roi_pool fwd + bwd, resize fwd / channels_last / bwd in a random mode, rotated IoU); rotated pairs on which the "
      f"reference's own float32 arithmetic is unstable (accepted when equal to a value the reference gives within 2 ulps of the inputs): {unstable_pairs}")
