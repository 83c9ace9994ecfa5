"""Parity tests proper (`-m gpu`, MI355X): every HIP kernel, called through the dispatcher
glue that sits directly on the C ABI, against
  * the plain-C oracle (oracle/tvmi_oracle.c) on seeded inputs,
  * the committed golden vectors (tests/golden, generated from the reference's CPU kernels),
  * the reference CPU kernels themselves (oracle/_ref) when the prebuilt library travelled.
Bars: bit-exact for NMS index lists / argmax / channel maps; 1e-4 for fp32 values (most ops are
in fact identical); 5e-3 for bf16/fp16 against the fp32 result on rounded inputs."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import vision_amd
from oracle import oracle as O
from helpers import adversarial_nms_inputs, gen, golden, random_boxes, rois_for

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 1e-4  # fp32 bar stated by BASELINE.json's north_star


def t(a, dtype=None):
    x = torch.as_tensor(a)
    return x.to(DEV) if dtype is None else x.to(DEV, dtype)


def test_native_library_is_the_one_running():
    assert torch.cuda.is_available()
    maps = open("/proc/self/maps").read()
    assert "libtvmi_kernels.so" in maps and "tvmi_torch.so" in maps and "tvmi_torch_stable.so" in maps
    assert "gfx950" in torch.cuda.get_device_properties(0).gcnArchName


# ------------------------------------------------------------------------------ NMS
def test_nms_golden_bit_exact(tv):
    g = golden("nms")
    for i in range(int(g["count"])):
        keep = tv.nms(t(g[f"boxes{i}"]), t(g[f"scores{i}"]), float(g[f"thr{i}"]))
        assert np.array_equal(keep.cpu().numpy(), g[f"keep{i}"]), f"case {i}"


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 127, 128, 129, 511, 513, 1000, 4097])
def test_nms_sizes_vs_oracle(tv, n):
    for seed, thr in ((0, 0.5), (1, 0.2), (2, 0.8)):
        boxes, scores = adversarial_nms_inputs(n, thr, gen(seed), dup=(seed == 2)) if n > 1 else (
            torch.tensor([[0.0, 0.0, 1.0, 1.0]]), torch.tensor([0.3]))
        keep = tv.nms(boxes.to(DEV), scores.to(DEV), thr)
        assert keep.dtype == torch.int64
        assert np.array_equal(keep.cpu().numpy(), O.nms(boxes.numpy(), scores.numpy(), thr))


def test_nms_dtypes_and_edges(tv):
    b, s = adversarial_nms_inputs(300, 0.5, gen(3))
    ref = O.nms(b.double().numpy(), s.double().numpy(), 0.5)
    assert np.array_equal(tv.nms(b.double().to(DEV), s.double().to(DEV), 0.5).cpu().numpy(), ref)
    # fp16 known-answer boxes of the reference (test/test_ops.py:1009-1022): same result as fp32
    kb = torch.tensor([[285.3538, 185.5758, 1193.5110, 851.4551], [285.1472, 188.7374, 1192.4984, 851.0669],
                       [279.2440, 197.9812, 1189.4746, 849.2019]])
    ks = torch.tensor([0.6370, 0.7569, 0.3966])
    k32 = tv.nms(kb.to(DEV), ks.to(DEV), 0.2)
    k16 = tv.nms(kb.to(DEV).half(), ks.to(DEV).half(), 0.2)
    assert torch.equal(k32, k16) and k32.tolist() == [1]
    # empty, degenerate, negative threshold
    assert tv.nms(torch.empty(0, 4, device=DEV), torch.empty(0, device=DEV), 0.5).shape == (0,)
    deg = torch.rand(100, 4, generator=gen(4)) * 20
    deg[:, 2:] = deg[:, :2]
    sc = torch.rand(100, generator=gen(5))
    for thr in (-1.0, 0.0, 0.5):
        assert np.array_equal(tv.nms(deg.to(DEV), sc.to(DEV), thr).cpu().numpy(), O.nms(deg.numpy(), sc.numpy(), thr))
    with pytest.raises(RuntimeError, match="boxes should be a 2d tensor"):
        tv.nms(torch.rand(4, device=DEV), torch.rand(4, device=DEV), 0.5)
    with pytest.raises(RuntimeError, match="boxes and scores should have same number of elements"):
        tv.nms(torch.rand(3, 4, device=DEV), torch.rand(4, device=DEV), 0.5)


def test_nms_against_reference_cpu_kernel(tv, need_ref):
    for seed in range(4):
        b, s = adversarial_nms_inputs(2500, 0.7, gen(10 + seed), dup=(seed % 2 == 1))
        assert torch.equal(tv.nms(b.to(DEV), s.to(DEV), 0.7).cpu(), tv.nms(b, s, 0.7))


def test_nms_threshold_edge_is_exact_without_the_division(tv):
    """The mask kernels decide most pairs with two float products and only divide inside a narrow band around the
    threshold; the outcome must equal the reference's `(double)(inter / union) > thr` for EVERY pair, in particular
    for IoUs that sit exactly on, one ulp above or one ulp below the threshold (float and double neighbours), for
    zero / negative unions and for thresholds outside the fast path's range."""
    g = gen(29)
    n = 1500
    # integer-coordinate boxes on a small canvas: thousands of pairs share exactly representable IoUs (1/2, 1/3, 2/3, ...)
    xy = torch.randint(0, 12, (n, 2), generator=g).float()
    wh = torch.randint(1, 7, (n, 2), generator=g).float()
    boxes = torch.cat([xy, xy + wh], 1)
    boxes[::97, 2:] = boxes[::97, :2]            # zero-area boxes: 0/0
    boxes[5::131, 2] = boxes[5::131, 0] - 2.0    # inverted boxes: negative area / union
    scores = torch.rand(n, generator=g)
    idxs = torch.randint(0, 3, (n,), generator=g)
    thrs = []
    for q in (0.5, 1.0 / 3.0, 2.0 / 3.0, 0.25, 0.2, 0.6):
        f = np.float32(q)
        thrs += [float(q), float(f), float(np.nextafter(f, np.float32(0))), float(np.nextafter(f, np.float32(1))),
                 float(np.nextafter(np.float64(f), 0.0)), float(np.nextafter(np.float64(f), 1.0))]
    thrs += [0.0, 1e-40, -0.5, 1.0, 2.0, float("inf")]
    bd, sd, idd = boxes.to(DEV), scores.to(DEV), idxs.to(DEV)
    for thr in thrs:
        want = O.nms(boxes.numpy(), scores.numpy(), thr)
        assert np.array_equal(tv.nms(bd, sd, thr).cpu().numpy(), want), thr
        wseg = O.nms(boxes.numpy(), scores.numpy(), thr, idxs.numpy())
        assert np.array_equal(torch.ops.tvmi.nms_segmented(bd, sd, idd, thr, 3).cpu().numpy(), wseg), thr
    big = torch.cat([boxes] * 4) + torch.arange(4 * n)[:, None].float() * 0   # n > 4096: segment-major kernel
    sbig = torch.rand(4 * n, generator=g)
    ibig = torch.randint(0, 5, (4 * n,), generator=g)
    for thr in thrs[:12]:
        want = O.nms(big.numpy(), sbig.numpy(), thr, ibig.numpy())
        assert np.array_equal(torch.ops.tvmi.nms_segmented(big.to(DEV), sbig.to(DEV), ibig.to(DEV), thr, -1).cpu().numpy(), want), thr


def test_nms_area_range_of_the_fast_path(tv):
    """The division-free tile loop is only taken for tiles whose boxes all have an area in [2^-60, 2^60]; boxes scaled
    to 2^-70 / 2^+70 areas, NaN / inf coordinates, zero-area and inverted boxes, normalised (0..1) coordinates — alone
    and mixed into otherwise ordinary tiles — must give the reference's index lists (exact path or fast path alike)."""
    g = gen(31)
    n = 3000
    base = random_boxes(n, 64, 64, 2, 30, g)
    scores = torch.rand(n, generator=g)
    idxs = torch.randint(0, 4, (n,), generator=g)
    variants = {"normalised": base / 64.0, "tiny": base * 2.0 ** -38, "huge": base * 2.0 ** 33, "edge_lo": base * 2.0 ** -32,
                "edge_hi": base * 2.0 ** 27}
    mixed = base.clone()
    mixed[7::61] *= 2.0 ** -38                   # a few out-of-range boxes per tile
    mixed[11::173] *= 2.0 ** 34
    mixed[13::211, 2] = float("nan")
    mixed[17::223, 3] = float("inf")
    mixed[19::227, 0] = float("-inf")
    mixed[23::97, 2:] = mixed[23::97, :2]        # zero area
    mixed[29::131, 2] = mixed[29::131, 0] - 1.0  # inverted
    variants["mixed"] = mixed
    for name, b in variants.items():
        for thr in (0.5, 0.3):
            want = O.nms(b.numpy(), scores.numpy(), thr)
            assert np.array_equal(tv.nms(b.to(DEV), scores.to(DEV), thr).cpu().numpy(), want), (name, thr)
            wseg = O.nms(b.numpy(), scores.numpy(), thr, idxs.numpy())
            got = torch.ops.tvmi.nms_segmented(b.to(DEV), scores.to(DEV), idxs.to(DEV), thr, 4).cpu().numpy()
            assert np.array_equal(got, wseg), (name, thr)
    big = torch.cat([mixed, base, mixed * 0.5])  # n > 4096: chunked mask kernels / segment-major kernel
    sbig = torch.rand(big.shape[0], generator=g)
    ibig = torch.randint(0, 6, (big.shape[0],), generator=g)
    assert np.array_equal(tv.nms(big.to(DEV), sbig.to(DEV), 0.5).cpu().numpy(), O.nms(big.numpy(), sbig.numpy(), 0.5))
    assert np.array_equal(torch.ops.tvmi.nms_segmented(big.to(DEV), sbig.to(DEV), ibig.to(DEV), 0.5, -1).cpu().numpy(),
                          O.nms(big.numpy(), sbig.numpy(), 0.5, ibig.numpy()))


def test_batched_nms_follows_the_reference_switch_of_arithmetic():
    """ops/boxes.py:83-109: up to 100,000 elements the reference evaluates the IoUs on boxes shifted by
    idxs * (max + 1); fp32 rounding of the shifted coordinates flips pairs at the threshold edge.  Seed 9 is such a
    case (the shifted and the per-category results differ by one box on the CPU): the mirror must give the
    reference's answer in both regimes."""
    g = gen(9)
    n = 20000
    boxes = random_boxes(n, 1000, 1000, 1, 101, g)
    scores = torch.rand(n, generator=g)
    idxs = torch.randint(0, 80, (n,), generator=g)
    want_trick = O.batched_nms(boxes, scores, idxs, 0.5)
    want_loop = O.nms(boxes.numpy(), scores.numpy(), 0.5, idxs.numpy())
    assert not np.array_equal(want_trick, want_loop)           # the case discriminates
    keep = vision_amd.batched_nms(boxes.to(DEV), scores.to(DEV), idxs.to(DEV), 0.5).cpu().numpy()
    assert np.array_equal(keep, want_trick)
    keep = torch.ops.tvmi.nms_segmented(boxes.to(DEV), scores.to(DEV), idxs.to(DEV), 0.5, -1).cpu().numpy()
    assert np.array_equal(keep, want_loop)                     # the schema-free op keeps the per-category arithmetic


def test_batched_nms_native_segmented_path():
    g = gen(6)
    n = 30000  # numel 120k > 100k -> "vanilla" semantics via tvmi::nms_segmented
    boxes = random_boxes(n, 1000, 1000, 1, 101, g)
    scores = torch.rand(n, generator=g)
    idxs = torch.randint(0, 80, (n,), generator=g)
    keep = vision_amd.batched_nms(boxes.to(DEV), scores.to(DEV), idxs.to(DEV), 0.5).cpu().numpy()
    assert np.array_equal(keep, O.nms(boxes.numpy(), scores.numpy(), 0.5, idxs.numpy()))
    # small input -> coordinate trick; both strategies agree (reference test_batched_nms_implementations)
    small = slice(0, 1000)
    a = vision_amd.boxes._batched_nms_vanilla(boxes[small].to(DEV), scores[small].to(DEV), idxs[small].to(DEV), 0.5)
    b = vision_amd.boxes._batched_nms_coordinate_trick(boxes[small].to(DEV), scores[small].to(DEV), idxs[small].to(DEV), 0.5)
    assert torch.equal(a, b)


def test_batched_nms_segment_major_layouts():
    """Segment-major path (n > 4096): ragged segments — singletons, segments that start and end inside one 64-box
    block, segments spanning many blocks, duplicate scores (stable tie order), sparse / negative / huge ids — and a
    segment above the 8,192-box limit, which must fall back to the global-order pipeline with the same answer."""
    g = gen(61)
    n = 12000
    boxes = random_boxes(n, 600, 600, 4, 120, g)
    scores = (torch.rand(n, generator=g) * 64).floor() / 64          # many exact ties
    sizes = [1, 1, 2, 63, 64, 65, 1, 700, 3, 1500, 5000]
    idxs = torch.cat([torch.full((m,), i, dtype=torch.int64) for i, m in enumerate(sizes)])
    idxs = torch.cat([idxs, torch.randint(100, 140, (n - len(idxs),), generator=g)])
    remap = torch.arange(200) * 7919 - 300_000                         # sparse ids, some negative
    remap[3] = 2 ** 40
    idxs = remap[idxs][torch.randperm(n, generator=g)]
    keep = torch.ops.tvmi.nms_segmented(boxes.to(DEV), scores.to(DEV), idxs.to(DEV), 0.4).cpu().numpy()
    assert np.array_equal(keep, O.nms(boxes.numpy(), scores.numpy(), 0.4, idxs.numpy()))
    big = torch.where(torch.arange(n) < 9000, torch.zeros(n, dtype=torch.int64), idxs)  # one 9000-box segment
    keep = torch.ops.tvmi.nms_segmented(boxes.to(DEV), scores.to(DEV), big.to(DEV), 0.4).cpu().numpy()
    assert np.array_equal(keep, O.nms(boxes.numpy(), scores.numpy(), 0.4, big.numpy()))
    # float64 boxes through the same path
    keep = torch.ops.tvmi.nms_segmented(boxes.double().to(DEV), scores.double().to(DEV), idxs.to(DEV), 0.4).cpu().numpy()
    assert np.array_equal(keep, O.nms(boxes.double().numpy(), scores.double().numpy(), 0.4, idxs.numpy()))


def test_batched_nms_single_launch_small_path():
    """n <= 4096 with a promised id range: the small path (per-segment tiles, concurrent sweeps, last-workgroup compaction)
    must equal the oracle bit for bit — ragged segments, empty ids, ties, fp64 — and inputs that break its limits
    (a segment above 1024 boxes, an id outside the promised range) must still give the right answer (fallback)."""
    g = gen(63)
    for n, S, thr in ((4000, 4, 0.5), (4096, 91, 0.3), (777, 1024, 0.7), (65, 3, 0.5), (1, 1, 0.5)):
        boxes = random_boxes(n, 500, 400, 4, 150, g)
        scores = (torch.rand(n, generator=g) * 128).floor() / 128
        idxs = torch.randint(0, S, (n,), generator=g)
        idxs[idxs == 1] = 0                                              # id 1 stays empty
        keep = vision_amd.batched_nms(boxes.to(DEV), scores.to(DEV), idxs.to(DEV), thr, num_segments=S).cpu().numpy()
        assert np.array_equal(keep, O.batched_nms(boxes, scores, idxs, thr)), (n, S)
    keep = vision_amd.batched_nms(boxes.double().to(DEV), scores.double().to(DEV), idxs.to(DEV), 0.5, num_segments=1).cpu().numpy()
    assert np.array_equal(keep, O.batched_nms(boxes.double(), scores.double(), idxs, 0.5))
    n = 3000
    boxes = random_boxes(n, 500, 400, 4, 150, g)
    scores = torch.rand(n, generator=g)
    big = torch.where(torch.arange(n) < 1500, torch.zeros(n, dtype=torch.int64), torch.randint(1, 5, (n,), generator=g))
    keep = vision_amd.batched_nms(boxes.to(DEV), scores.to(DEV), big.to(DEV), 0.5, num_segments=5).cpu().numpy()
    assert np.array_equal(keep, O.batched_nms(boxes, scores, big, 0.5))                             # 1500-box segment
    keep = vision_amd.batched_nms(boxes.to(DEV), scores.to(DEV), (big + 7).to(DEV), 0.5, num_segments=5).cpu().numpy()
    assert np.array_equal(keep, O.batched_nms(boxes, scores, big + 7, 0.5))                         # ids outside [0, 5)
    _, num = vision_amd.boxes.batched_nms_padded(boxes.to(DEV), scores.to(DEV), big.to(DEV), 0.5, num_segments=5)
    assert int(num) == -1                                                                           # sync-free form reports it


def test_nms_large_path_dependency_chains(tv):
    """n > 4096 (chunked mask kernels, sweep on a second stream, parallel fixed-point rounds per 1024-box group): a
    chain of boxes in which every box is suppressed only by its predecessor (keep, drop, keep, drop, ...) is as deep as
    a dependency gets — far beyond the fixed-point round budget, so the serial walk has to take over — mixed with
    ordinary random boxes; index lists must equal the oracle's."""
    g = gen(97)
    n_chain, n_rand = 5000, 3000
    x = torch.arange(n_chain, dtype=torch.float32) * 4.0                    # 10-wide boxes every 4 px: IoU(i, i+1) = 6/14
    chain = torch.stack([x, torch.zeros(n_chain), x + 10.0, torch.full((n_chain,), 10.0)], 1)
    rand = random_boxes(n_rand, 20000, 400, 5, 60, g)
    rand[:, 1] += 50.0                                                      # keep them off the chain's row
    rand[:, 3] += 50.0
    boxes = torch.cat([chain, rand])
    scores = torch.cat([torch.linspace(1.0, 0.5, n_chain), torch.rand(n_rand, generator=g) * 0.4])
    perm = torch.randperm(n_chain + n_rand, generator=g)
    boxes, scores = boxes[perm], scores[perm]
    for thr in (0.4, 0.45):
        keep = tv.nms(boxes.to(DEV), scores.to(DEV), thr).cpu().numpy()
        want = O.nms(boxes.numpy(), scores.numpy(), thr)
        assert np.array_equal(keep, want), thr
    assert len(want) > n_chain // 2


def test_nms_100k_properties(tv):
    # BASELINE config 3 size; oracle too slow here -> size-independent properties
    g = gen(7)
    n = 100_000
    boxes = random_boxes(n, 1000, 1000, 1, 101, g).to(DEV)
    scores = torch.rand(n, generator=g).to(DEV)
    keep = tv.nms(boxes, scores, 0.5)
    ks = scores[keep]
    assert torch.all(ks[:-1] >= ks[1:])                       # score order
    assert keep.unique().numel() == keep.numel()              # no duplicates
    kb = boxes[keep]
    # idempotence: NMS of the kept set keeps everything
    assert tv.nms(kb, ks, 0.5).numel() == keep.numel()
    # every suppressed box overlaps (> thr) some kept box with a higher score (check a sample)
    mask = torch.ones(n, dtype=torch.bool, device=DEV)
    mask[keep] = False
    sup = torch.nonzero(mask)[:, 0][:2000]
    iou = vision_amd.box_iou(boxes[sup], kb)
    higher = ks[None, :] >= scores[sup][:, None]
    assert torch.all(((iou > 0.5) & higher).any(dim=1))
    # prefix property: the first 4096 boxes by score processed alone give the same decisions
    order = scores.argsort(descending=True, stable=True)[:4096]
    sub = tv.nms(boxes[order], scores[order], 0.5)
    assert torch.equal(order[sub], keep[: sub.numel()])


# ------------------------------------------------------------------------------ RoI family
def test_roi_ops_golden(tv):
    g = golden("roi_ops")
    x, rois = t(g["x"]), t(g["rois"])
    C = x.shape[1]
    for scale in (1.0, 0.5):
        for sr in (-1, 2):
            for aligned in (False, True):
                key = f"s{scale}_sr{sr}_a{int(aligned)}"
                y = tv.roi_align(x, rois, scale, 5, 5, sr, aligned)
                np.testing.assert_allclose(y.cpu().numpy(), g["roi_align_" + key], rtol=0, atol=1e-6)
                gr = torch.linspace(-1, 1, y.numel()).reshape(y.shape).to(DEV)
                gin = tv._roi_align_backward(gr, rois, scale, 5, 5, 2, C, 10, 10, sr, aligned)
                np.testing.assert_allclose(gin.cpu().numpy(), g["roi_align_bwd_" + key], rtol=0, atol=1e-5)
            y, m = tv.ps_roi_align(x, rois[:-1], scale, 5, 5, sr)
            np.testing.assert_allclose(y.cpu().numpy(), g[f"ps_roi_align_s{scale}_sr{sr}"], rtol=0, atol=1e-6, equal_nan=True)
            assert np.array_equal(m.cpu().numpy(), g[f"ps_roi_align_map_s{scale}_sr{sr}"])
        y, a = tv.roi_pool(x, rois, scale, 5, 5)
        assert np.array_equal(y.cpu().numpy(), g[f"roi_pool_s{scale}"]) and np.array_equal(a.cpu().numpy(), g[f"roi_pool_argmax_s{scale}"])
        y, m = tv.ps_roi_pool(x, rois, scale, 5, 5)
        np.testing.assert_allclose(y.cpu().numpy(), g[f"ps_roi_pool_s{scale}"], rtol=0, atol=1e-6)
        assert np.array_equal(m.cpu().numpy(), g[f"ps_roi_pool_map_s{scale}"])


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("contiguous", [True, False])
def test_roi_align_reference_style(tv, dtype, contiguous):
    # shape of the reference's RoIOpTester.test_forward (test/test_ops.py:127-163)
    g = gen(8)
    x = torch.rand(2, 50, 10, 10, generator=g).to(dtype)
    if not contiguous:
        x = x.permute(0, 1, 3, 2)
    rois = torch.tensor([[0, 0, 0, 9, 9], [0, 0, 5, 4, 9], [0, 5, 5, 9, 9], [1, 0, 0, 9, 9]], dtype=dtype)
    tol = {torch.float32: 1e-5, torch.float64: 1e-9, torch.float16: 4e-3, torch.bfloat16: 5e-3}[dtype]
    for aligned in (False, True):
        for sr in (-1, 2):
            y = tv.roi_align(x.to(DEV), rois.to(DEV), 1.0, 5, 5, sr, aligned)
            assert y.dtype == dtype
            odt = np.float64 if dtype == torch.float64 else np.float32
            ref = O.roi_align(x.contiguous().to(torch.float64 if dtype == torch.float64 else torch.float32).numpy().astype(odt),
                              rois.to(torch.float64).numpy().astype(odt), 1.0, 5, 5, sr, aligned)
            np.testing.assert_allclose(y.float().cpu().numpy() if dtype != torch.float64 else y.cpu().numpy(), ref, rtol=tol, atol=tol)


@pytest.mark.parametrize("ph,pw,sr,aligned", [(7, 7, 2, False), (14, 14, 2, False), (7, 7, 0, True), (3, 5, 3, True),
                                               (1, 1, 1, False), (7, 7, 2, True), (2, 9, 0, False)])
def test_roi_align_fpn_shapes_fwd_bwd(tv, ph, pw, sr, aligned):
    g = gen(9)
    N, C, H, W = 2, 40, 50, 84
    x = torch.randn(N, C, H, W, generator=g)
    rois = rois_for(N, 150, W * 16, H * 16, 16, 600, g)
    rois[0, 1:] = torch.tensor([-50.0, -30.0, 2000.0, 1500.0])   # larger than the map (LDS fallback path)
    rois[1, 1:] = torch.tensor([100.0, 100.0, 100.0, 100.0])     # zero-size
    rois[2, 1:] = torch.tensor([5000.0, 5000.0, 6000.0, 6000.0]) # fully outside
    rois[3, 1:] = torch.tensor([600.0, 500.0, 100.0, 80.0])      # malformed (x2 < x1): negative bins when aligned
    rois[4, 1:] = torch.tensor([1300.0, 700.0, 1343.0, 799.0])    # touches the bottom-right corner
    scale = 1 / 16
    y = tv.roi_align(x.to(DEV), rois.to(DEV), scale, ph, pw, sr, aligned)
    ref = O.roi_align(x.numpy(), rois.numpy(), scale, ph, pw, sr, aligned)
    np.testing.assert_allclose(y.cpu().numpy(), ref, rtol=0, atol=TOL)
    gr = torch.randn(ref.shape, generator=g)
    gin = tv._roi_align_backward(gr.to(DEV), rois.to(DEV), scale, ph, pw, N, C, H, W, sr, aligned)
    refb = O.roi_align_backward(gr.numpy(), rois.numpy(), scale, ph, pw, N, C, H, W, sr, aligned)
    np.testing.assert_allclose(gin.cpu().numpy(), refb, rtol=1e-4, atol=TOL * max(1.0, float(np.abs(refb).max())) )
    # non-contiguous grad (strides honoured, cpu/roi_align_kernel.cpp:370-373)
    gr_nc = gr.permute(0, 1, 3, 2).contiguous().permute(0, 1, 3, 2)
    gin2 = tv._roi_align_backward(gr_nc.to(DEV), rois.to(DEV), scale, ph, pw, N, C, H, W, sr, aligned)
    np.testing.assert_allclose(gin2.cpu().numpy(), refb, rtol=1e-4, atol=TOL * max(1.0, float(np.abs(refb).max())))


def test_roi_align_large_window_on_stride4_map(tv):
    # RoIs covering most of a stride-4 map do not fit the LDS window -> table-driven global gathers
    g = gen(10)
    x = torch.randn(1, 8, 200, 272, generator=g)
    rois = torch.tensor([[0, 0.0, 0.0, 1087.0, 799.0], [0, 10.0, 20.0, 900.0, 700.0], [0, 500.0, 300.0, 520.0, 330.0]])
    for ph, sr in ((7, 2), (14, 2), (7, 0)):
        y = tv.roi_align(x.to(DEV), rois.to(DEV), 0.25, ph, ph, sr, False)
        np.testing.assert_allclose(y.cpu().numpy(), O.roi_align(x.numpy(), rois.numpy(), 0.25, ph, ph, sr, False), rtol=0, atol=TOL)


def test_roi_align_empty_and_errors(tv):
    x = torch.rand(1, 3, 8, 8, device=DEV)
    assert tv.roi_align(x, torch.empty(0, 5, device=DEV), 1.0, 2, 2, 2, False).shape == (0, 3, 2, 2)
    assert tv._roi_align_backward(torch.empty(0, 3, 2, 2, device=DEV), torch.empty(0, 5, device=DEV), 1.0, 2, 2, 1, 3, 8, 8, 2,
                                  False).abs().sum().item() == 0
    with pytest.raises(RuntimeError, match="rois must have shape as Tensor\\[K, 5\\]"):
        tv.roi_align(x, torch.rand(2, 4, device=DEV), 1.0, 2, 2, 2, False)
    with pytest.raises(RuntimeError, match="same type"):
        tv.roi_align(x, torch.rand(2, 5, device=DEV).double(), 1.0, 2, 2, 2, False)


def test_roi_pool_and_ps_ops_vs_oracle(tv):
    g = gen(11)
    for dt in (torch.float32, torch.float64):
        x = torch.randn(2, 2 * 9, 17, 23, generator=g).to(dt)
        rois = rois_for(2, 60, 46, 34, 1, 30, g, dt)
        rois[0, 1:] = torch.tensor([-10.0, -10.0, 80.0, 70.0], dtype=dt)
        for scale in (0.5, 1.0):
            y, a = tv.roi_pool(x.to(DEV), rois.to(DEV), scale, 3, 3)
            ry, ra = O.roi_pool(x.numpy(), rois.numpy(), scale, 3, 3)
            assert np.array_equal(y.cpu().numpy(), ry) and np.array_equal(a.cpu().numpy(), ra)
            gr = torch.randn(ry.shape, generator=g).to(dt)
            gi = tv._roi_pool_backward(gr.to(DEV), rois.to(DEV), a, scale, 3, 3, 2, 18, 17, 23)
            np.testing.assert_allclose(gi.cpu().numpy(), O.roi_pool_backward(gr.numpy(), rois.numpy(), ra, 2, 18, 17, 23), atol=1e-4)
            for sr in (2, 0):
                y, m = tv.ps_roi_align(x.to(DEV), rois.to(DEV), scale, 3, 3, sr)
                ry, rm = O.ps_roi_align(x.numpy(), rois.numpy(), scale, 3, 3, sr)
                np.testing.assert_allclose(y.cpu().numpy(), ry, rtol=0, atol=1e-5, equal_nan=True)
                assert np.array_equal(m.cpu().numpy(), rm)
                gr2 = torch.randn(ry.shape, generator=g).to(dt)
                gi = tv._ps_roi_align_backward(gr2.to(DEV), rois.to(DEV), m, scale, 3, 3, sr, 2, 18, 17, 23)
                np.testing.assert_allclose(gi.cpu().numpy(), O.ps_roi_align_backward(gr2.numpy(), rois.numpy(), rm, scale, 3, 3, sr, 2, 18, 17, 23),
                                           atol=1e-4, equal_nan=True)
            y, m = tv.ps_roi_pool(x.to(DEV), rois.to(DEV), scale, 3, 3)
            ry, rm = O.ps_roi_pool(x.numpy(), rois.numpy(), scale, 3, 3)
            np.testing.assert_allclose(y.cpu().numpy(), ry, rtol=0, atol=1e-5)
            assert np.array_equal(m.cpu().numpy(), rm)
            gi = tv._ps_roi_pool_backward(gr2.to(DEV), rois.to(DEV), m, scale, 3, 3, 2, 18, 17, 23)
            np.testing.assert_allclose(gi.cpu().numpy(), O.ps_roi_pool_backward(gr2.numpy(), rois.numpy(), rm, scale, 3, 3, 2, 18, 17, 23), atol=1e-4)
    with pytest.raises(RuntimeError, match="input channels must be a multiple of pooling height \\* pooling width"):
        tv.ps_roi_align(torch.rand(1, 10, 8, 8, device=DEV), torch.rand(2, 5, device=DEV), 1.0, 3, 3, 2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
def test_roi_pool_wave_kernel_windows_and_bin_counts(tv, dtype):
    """Column-lane RoIPool kernel (7x7) and the lane-per-output kernel (other shapes): windows wider than a wave (64
    columns: scanned bin by bin), RoIs sticking out of the map, empty bins, equal maxima (first one in (h, w) order
    wins: argmax must match) — value and argmax identical to the reference arithmetic (16-bit inputs: on the rounded
    values)."""
    g = gen(51)
    N, C, H, W = 2, 45, 60, 90
    x = (torch.randn(N, C, H, W, generator=g) * 4).round().to(dtype)        # many exact ties
    rois = rois_for(N, 200, W * 8, H * 8, 4, 300, g)
    rois[0, 1:] = torch.tensor([0.0, 0.0, W * 8.0, H * 8.0])                # whole map: 5400-pixel window
    rois[1, 1:] = torch.tensor([-100.0, -80.0, 900.0, 600.0])
    rois[2, 1:] = torch.tensor([300.0, 200.0, 300.0, 200.0])                # one pixel
    rois[3, 1:] = torch.tensor([5000.0, 5000.0, 6000.0, 6000.0])            # outside: empty bins
    rois[4, 1:] = torch.tensor([400.0, 300.0, 100.0, 50.0])                 # malformed
    for P in (7, 9, 2):
        y, a = tv.roi_pool(x.to(DEV), rois.to(dtype).to(DEV), 0.125, P, P)
        ry, ra = O.roi_pool(x.float().numpy(), rois.to(dtype).float().numpy(), 0.125, P, P)
        assert y.dtype == dtype
        assert np.array_equal(y.float().cpu().numpy(), ry) and np.array_equal(a.cpu().numpy(), ra), P


@pytest.mark.parametrize("C", [7, 33, 70, 130, 256, 300])
def test_roi_pool_column_kernel_channel_partitions(tv, C):
    """The 7x7 kernel deals (RoI, 32-channel chunk) units to 8 partitions (chunk = partition mod 8 when there are 8 or
    more chunks, RoI slices when chunks divides 8, plain order otherwise): every channel count class, partial last
    chunks, narrow windows (2 / 4 / 8 channels side by side in a wave), -inf / NaN / -FLT_MAX pixels."""
    g = gen(60 + C)
    N, H, W = 3, 40, 56
    x = (torch.randn(N, C, H, W, generator=g) * 3).round()
    x[:, ::5, ::7, ::3] = float("-inf")
    x[:, 1::6, 3::11, 1::5] = float("nan")
    x[:, 2::9, :, :] = -3.4028234663852886e38                                    # whole planes at -FLT_MAX: argmax -1
    rois = rois_for(N, 150, W * 4, H * 4, 2, 200, g)
    rois[:20, 3] = rois[:20, 1] + torch.rand(20, generator=g) * 24                # narrow windows: <= 8 columns
    rois[20:40, 3] = rois[20:40, 1] + 30 + torch.rand(20, generator=g) * 30       # 8..16 columns
    y, a = tv.roi_pool(x.to(DEV), rois.to(DEV), 0.25, 7, 7)
    ry, ra = O.roi_pool(x.numpy(), rois.numpy(), 0.25, 7, 7)
    assert np.array_equal(a.cpu().numpy(), ra)
    assert np.array_equal(y.cpu().numpy(), ry, equal_nan=True)


def test_roi_pool_backward_plane_owner(tv):
    """RoIPool backward, plane-owner regime (one wave per gradient plane, LDS accumulation in program order): equals the
    oracle, is bit-reproducible, handles > 1024 RoIs per image-list chunk, RoIs in random image order, strided grads
    and 16-bit grads (fp32 accumulation); planes too large for ONE LDS plane are cut into row strips with one owner wave each
    (38,000 and 67,200 pixels: 2 strips; a 200x336 FPN level) and stay deterministic."""
    g = gen(81)
    for (N, C, H, W, K) in ((3, 37, 40, 56, 2500), (2, 5, 200, 190, 300), (1, 6, 200, 336, 500)):
        x = (torch.randn(N, C, H, W, generator=g) * 3).round()
        rois = rois_for(N, K, W * 4, H * 4, 4, 120, g)
        y, a = tv.roi_pool(x.to(DEV), rois.to(DEV), 0.25, 7, 7)
        gr = torch.randn(K, C, 7, 7, generator=g)
        want = O.roi_pool_backward(gr.numpy(), rois.numpy(), a.cpu().numpy(), N, C, H, W)
        got = tv._roi_pool_backward(gr.to(DEV), rois.to(DEV), a, 0.25, 7, 7, N, C, H, W)
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-4, atol=1e-4 * max(1.0, float(np.abs(want).max())))
        again = tv._roi_pool_backward(gr.to(DEV), rois.to(DEV), a, 0.25, 7, 7, N, C, H, W)
        assert torch.equal(got, again)                                       # fixed summation order (strips included)
        # strided grad (channels_last storage of [K, C, 7, 7]) and bf16
        gs = gr.to(DEV).contiguous(memory_format=torch.channels_last)
        got_s = tv._roi_pool_backward(gs, rois.to(DEV), a, 0.25, 7, 7, N, C, H, W)
        np.testing.assert_allclose(got_s.cpu().numpy(), want, rtol=1e-4, atol=1e-4 * max(1.0, float(np.abs(want).max())))
        g16 = gr.to(torch.bfloat16)
        got16 = tv._roi_pool_backward(g16.to(DEV), rois.to(torch.bfloat16).to(DEV), a, 0.25, 7, 7, N, C, H, W)
        want16 = O.roi_pool_backward(g16.float().numpy(), rois.numpy(), a.cpu().numpy(), N, C, H, W)
        tol16 = 2e-2 * max(1.0, float(np.abs(want16).max()))
        np.testing.assert_allclose(got16.float().cpu().numpy(), want16, rtol=0, atol=tol16)
    # deterministic flag: the plane regime does not raise under torch.use_deterministic_algorithms
    torch.use_deterministic_algorithms(True)
    try:
        x = torch.randn(1, 4, 20, 20, generator=g)
        rois = rois_for(1, 30, 80, 80, 4, 60, g)
        _, a = tv.roi_pool(x.to(DEV), rois.to(DEV), 0.25, 7, 7)
        tv._roi_pool_backward(torch.randn(30, 4, 7, 7, generator=g).to(DEV), rois.to(DEV), a, 0.25, 7, 7, 1, 4, 20, 20)
    finally:
        torch.use_deterministic_algorithms(False)


@pytest.mark.parametrize("align", [True, False])
def test_ps_roi_backward_plane_owner(tv, align):
    """PSRoIAlign / PSRoIPool backward, plane-owner regime: equals the oracle, bit-reproducible, > 64 and > 1024 RoIs per
    image, adaptive sampling grid (sampling_ratio 0), RoIs sticking out of the map; large planes take the atomic kernels."""
    g = gen(83 + int(align))
    for (N, C_out, H, W, K, P, sr) in ((2, 3, 30, 44, 2300, 3, 2), (3, 2, 25, 31, 150, 7, 0), (1, 2, 200, 190, 60, 3, 2)):
        C = C_out * P * P
        x = torch.randn(N, C, H, W, generator=g)
        rois = rois_for(N, K, W * 4, H * 4, 4, 100, g)
        rois[::7, 1:3] -= 20.0
        xd = x.to(DEV)
        if align:
            y, m = tv.ps_roi_align(xd, rois.to(DEV), 0.25, P, P, sr)
        else:
            y, m = tv.ps_roi_pool(xd, rois.to(DEV), 0.25, P, P)
        gr = torch.randn(K, C_out, P, P, generator=g)
        if align:
            got = tv._ps_roi_align_backward(gr.to(DEV), rois.to(DEV), m, 0.25, P, P, sr, N, C, H, W)
            again = tv._ps_roi_align_backward(gr.to(DEV), rois.to(DEV), m, 0.25, P, P, sr, N, C, H, W)
            want = O.ps_roi_align_backward(gr.numpy(), rois.numpy(), m.cpu().numpy(), 0.25, P, P, sr, N, C, H, W)
        else:
            got = tv._ps_roi_pool_backward(gr.to(DEV), rois.to(DEV), m, 0.25, P, P, N, C, H, W)
            again = tv._ps_roi_pool_backward(gr.to(DEV), rois.to(DEV), m, 0.25, P, P, N, C, H, W)
            want = O.ps_roi_pool_backward(gr.numpy(), rois.numpy(), m.cpu().numpy(), 0.25, P, P, N, C, H, W)
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-4, atol=1e-4 * max(1.0, float(np.abs(want).max())))
        if H * W <= 36864:
            assert torch.equal(got, again)


def test_roi_ops_autograd_on_gpu(tv):
    g = gen(12)
    x = torch.rand(1, 8, 9, 9, generator=g, dtype=torch.float64).to(DEV).requires_grad_(True)
    rois = torch.tensor([[0, 0.5, 1.0, 7.2, 8.0], [0, 2.0, 2.0, 5.0, 4.5]], dtype=torch.float64, device=DEV)
    assert torch.autograd.gradcheck(lambda v: vision_amd.roi_align(v, rois, 2, 1.0, 2, True), (x,), nondet_tol=1e-5)
    assert torch.autograd.gradcheck(lambda v: vision_amd.roi_align(v, rois, 3, 0.7, -1, False), (x,), nondet_tol=1e-5)
    assert torch.autograd.gradcheck(lambda v: vision_amd.ps_roi_align(v, rois, 2, 1.0, 2), (x,), nondet_tol=1e-5)
    assert torch.autograd.gradcheck(lambda v: vision_amd.ps_roi_pool(v, rois, 2, 1.0), (x,), nondet_tol=1e-5)
    assert torch.autograd.gradcheck(lambda v: vision_amd.roi_pool(v, rois, 2, 1.0), (x,), nondet_tol=1e-5)
    with torch.autocast("cuda", dtype=torch.float16):
        y = vision_amd.roi_align(x.detach().half(), rois.half(), 2, 1.0, 2, False)
    assert y.dtype == torch.float16


def test_multiscale_roi_align_config2_shapes(tv):
    # BASELINE config 2 at reduced channel count: 4 FPN levels of a padded 800x1344 batch
    g = gen(13)
    B, C = 2, 16
    feats = {str(i): torch.randn(B, C, 800 // s, 1344 // s, generator=g) for i, s in enumerate((4, 8, 16, 32))}
    boxes = [random_boxes(300, 1344, 800, 8, 700, g) for _ in range(B)]
    pool = vision_amd.MultiScaleRoIAlign(["0", "1", "2", "3"], 7, 2)
    out = pool({k: v.to(DEV) for k, v in feats.items()}, [b.to(DEV) for b in boxes], [(800, 1344)] * B).cpu()
    lv = pool.map_levels(boxes)
    rois = torch.cat([torch.cat([torch.full((len(b), 1), float(i)), b], 1) for i, b in enumerate(boxes)])
    for lvl in range(4):
        sel = torch.nonzero(lv == lvl)[:, 0]
        if sel.numel() == 0:
            continue
        ref = O.roi_align(feats[str(lvl)].numpy(), rois[sel].numpy(), pool.scales[lvl], 7, 7, 2, False)
        np.testing.assert_allclose(out[sel].numpy(), ref, rtol=0, atol=TOL)


# ------------------------------------------------------------------------------ deform_conv2d
def test_deform_conv2d_golden(tv):
    g = golden("deform_conv2d")
    x, w, off, m, b = (t(g[k]) for k in ("x", "weight", "offset", "mask", "bias"))
    args = (2, 1, 1, 0, 2, 1, 2, 3)
    np.testing.assert_allclose(tv.deform_conv2d(x, w, off, m, b, *args, True).cpu().numpy(), g["out_mask"], rtol=0, atol=TOL)
    np.testing.assert_allclose(tv.deform_conv2d(x, w, off, torch.zeros(4, 1, device=DEV), b, *args, False).cpu().numpy(),
                               g["out_nomask"], rtol=0, atol=TOL)
    gr = torch.linspace(-1, 1, g["out_mask"].size).reshape(g["out_mask"].shape).to(DEV)
    grads = tv._deform_conv2d_backward(gr, x, w, off, m, b, *args, True)
    for nm, got in zip(("gin", "gw", "goff", "gmask", "gbias"), grads):
        np.testing.assert_allclose(got.cpu().numpy(), g["bwd_" + nm], rtol=1e-4, atol=TOL, err_msg=nm)


@pytest.mark.parametrize("cfg", [
    dict(B=2, C=64, OC=96, H=20, W=24, k=(3, 3), groups=1, og=2, stride=(1, 1), pad=(1, 1), dil=(1, 1), mask=True),   # MFMA 2x2
    dict(B=1, C=48, OC=200, H=13, W=17, k=(3, 3), groups=1, og=1, stride=(2, 1), pad=(1, 2), dil=(1, 2), mask=False), # MFMA 4x1
    dict(B=3, C=36, OC=40, H=11, W=9, k=(1, 3), groups=2, og=3, stride=(1, 1), pad=(0, 1), dil=(1, 1), mask=True),    # MFMA 1x4, og split
    dict(B=2, C=8, OC=8, H=9, W=9, k=(3, 3), groups=8, og=1, stride=(1, 1), pad=(1, 1), dil=(1, 1), mask=False),      # depthwise (direct)
    dict(B=2, C=6, OC=2, H=5, W=4, k=(3, 2), groups=2, og=3, stride=(2, 1), pad=(1, 0), dil=(2, 1), mask=True),       # reference test cfg
])
def test_deform_conv2d_vs_oracle(tv, cfg):
    g = gen(14)
    kh, kw = cfg["k"]
    oh = (cfg["H"] + 2 * cfg["pad"][0] - (cfg["dil"][0] * (kh - 1) + 1)) // cfg["stride"][0] + 1
    ow = (cfg["W"] + 2 * cfg["pad"][1] - (cfg["dil"][1] * (kw - 1) + 1)) // cfg["stride"][1] + 1
    x = torch.randn(cfg["B"], cfg["C"], cfg["H"], cfg["W"], generator=g)
    w = torch.randn(cfg["OC"], cfg["C"] // cfg["groups"], kh, kw, generator=g) * 0.1
    off = torch.randn(cfg["B"], 2 * cfg["og"] * kh * kw, oh, ow, generator=g) * 2
    m = torch.rand(cfg["B"], cfg["og"] * kh * kw, oh, ow, generator=g)
    b = torch.randn(cfg["OC"], generator=g)
    y = vision_amd.deform_conv2d(x.to(DEV), off.to(DEV), w.to(DEV), b.to(DEV), cfg["stride"], cfg["pad"], cfg["dil"],
                                 m.to(DEV) if cfg["mask"] else None)
    ref = O.deform_conv2d(x.numpy(), w.numpy(), off.numpy(), m.numpy(), b.numpy(), cfg["stride"], cfg["pad"], cfg["dil"],
                          cfg["groups"], cfg["og"], cfg["mask"])
    np.testing.assert_allclose(y.cpu().numpy(), ref, rtol=1e-4, atol=TOL)


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 5e-3), (torch.bfloat16, 3e-2)], ids=["fp16", "bf16"])
@pytest.mark.parametrize("cfg", [
    dict(B=2, C=64, OC=96, H=20, W=24, k=(3, 3), groups=1, og=2, stride=(1, 1), pad=(1, 1), dil=(1, 1), mask=True),   # 2x4 waves, slabs of 16
    dict(B=1, C=48, OC=200, H=13, W=17, k=(3, 3), groups=1, og=1, stride=(2, 1), pad=(1, 2), dil=(1, 2), mask=False), # 4x2 waves, OC tail
    dict(B=3, C=36, OC=40, H=11, W=9, k=(1, 3), groups=2, og=3, stride=(1, 1), pad=(0, 1), dil=(1, 1), mask=True),    # ICg 18, offset groups of 12: partial slabs
    dict(B=2, C=40, OC=24, H=9, W=31, k=(3, 3), groups=1, og=5, stride=(1, 1), pad=(1, 1), dil=(1, 1), mask=True),    # offset groups of 8
])
def test_deform_conv2d_16bit_mfma_kernel(cfg, dtype, tol):
    """fp16 / bf16 deform_conv2d on v_mfma_f32_32x32x16_{f16,bf16} (deform_conv2d.hip dcn_fwd_mfma_16): 16-deep K slabs in a
    [row][k] LDS layout, weight slabs that offset-group boundaries cut into pieces, OC / pixel tails — against the reference
    arithmetic in fp32 on the rounded tensors (oracle).  The reference rounds its sampled columns to the 16-bit type too and
    accumulates in 16 bits (cuda/deform_conv2d_kernel.cu:1234-1239); bar = a few 16-bit ulps of the output scale (~1)."""
    g = gen(19)
    kh, kw = cfg["k"]
    oh = (cfg["H"] + 2 * cfg["pad"][0] - (cfg["dil"][0] * (kh - 1) + 1)) // cfg["stride"][0] + 1
    ow = (cfg["W"] + 2 * cfg["pad"][1] - (cfg["dil"][1] * (kw - 1) + 1)) // cfg["stride"][1] + 1
    x = torch.randn(cfg["B"], cfg["C"], cfg["H"], cfg["W"], generator=g).to(dtype)
    w = (torch.randn(cfg["OC"], cfg["C"] // cfg["groups"], kh, kw, generator=g) * 0.1).to(dtype)
    off = (torch.randn(cfg["B"], 2 * cfg["og"] * kh * kw, oh, ow, generator=g) * 2).to(dtype)
    m = torch.rand(cfg["B"], cfg["og"] * kh * kw, oh, ow, generator=g).to(dtype)
    b = torch.randn(cfg["OC"], generator=g).to(dtype)
    y = vision_amd.deform_conv2d(x.to(DEV), off.to(DEV), w.to(DEV), b.to(DEV), cfg["stride"], cfg["pad"], cfg["dil"],
                                 m.to(DEV) if cfg["mask"] else None)
    assert y.dtype == dtype
    ref = O.deform_conv2d(x.float().numpy(), w.float().numpy(), off.float().numpy(), m.float().numpy(), b.float().numpy(),
                          cfg["stride"], cfg["pad"], cfg["dil"], cfg["groups"], cfg["og"], cfg["mask"])
    np.testing.assert_allclose(y.float().cpu().numpy(), ref, rtol=tol, atol=tol)


@pytest.mark.parametrize("cfg", [
    dict(B=2, C=40, H=37, W=150, og=1, stride=(1, 1), pad=(1, 1), dil=(1, 1), mask=True, sigma=1.0),     # several tiles, ragged edges
    dict(B=1, C=64, H=21, W=70, og=4, stride=(1, 1), pad=(1, 1), dil=(1, 1), mask=False, sigma=6.0),     # offsets far beyond the staged halo
    dict(B=2, C=24, H=30, W=41, og=3, stride=(2, 2), pad=(2, 1), dil=(2, 2), mask=True, sigma=2.0),      # stride / dilation / offset groups
    dict(B=1, C=17, H=9, W=5, og=1, stride=(1, 2), pad=(0, 3), dil=(1, 1), mask=True, sigma=3.0),        # tiny map, odd channel count
])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_deform_conv2d_depthwise_kernel(cfg, dtype):
    """groups == C == OC, 3x3: the lane = pixel LDS-window kernel (deform_conv2d.hip dcn_fwd_depthwise3x3) — taps inside the
    staged window read LDS, taps beyond the +/-4 px halo fall back to the reference arithmetic on global memory, zero
    padding comes from the zero-filled window border; channel ranges / chunks that do not divide evenly.  Reference:
    cpu/deform_conv2d_kernel.cpp:95-209 via the oracle; bf16 against fp32 on the rounded tensors at the 16-bit bar."""
    g = gen(18)
    C, og = cfg["C"], cfg["og"]
    if C % og:
        C = C // og * og
    oh = (cfg["H"] + 2 * cfg["pad"][0] - (cfg["dil"][0] * 2 + 1)) // cfg["stride"][0] + 1
    ow = (cfg["W"] + 2 * cfg["pad"][1] - (cfg["dil"][1] * 2 + 1)) // cfg["stride"][1] + 1
    x = torch.randn(cfg["B"], C, cfg["H"], cfg["W"], generator=g).to(dtype)
    w = (torch.randn(C, 1, 3, 3, generator=g) * 0.3).to(dtype)
    off = (torch.randn(cfg["B"], 2 * og * 9, oh, ow, generator=g) * cfg["sigma"]).to(dtype)
    m = torch.rand(cfg["B"], og * 9, oh, ow, generator=g).to(dtype)
    b = torch.randn(C, generator=g).to(dtype)
    y = vision_amd.deform_conv2d(x.to(DEV), off.to(DEV), w.to(DEV), b.to(DEV), cfg["stride"], cfg["pad"], cfg["dil"],
                                 m.to(DEV) if cfg["mask"] else None)
    assert y.dtype == dtype and tuple(y.shape) == (cfg["B"], C, oh, ow)
    ref = O.deform_conv2d(x.float().numpy(), w.float().numpy(), off.float().numpy(), m.float().numpy(), b.float().numpy(),
                          cfg["stride"], cfg["pad"], cfg["dil"], C, og, cfg["mask"])
    tol = TOL if dtype == torch.float32 else 2e-2
    np.testing.assert_allclose(y.float().cpu().numpy(), ref, rtol=tol, atol=tol)


DCN_BWD_CFGS = [
    dict(B=2, C=64, OC=96, H=20, W=24, k=(3, 3), groups=1, og=1, stride=(1, 1), pad=(1, 1), dil=(1, 1), mask=True),     # matrix cores
    dict(B=2, C=64, OC=96, H=20, W=24, k=(3, 3), groups=1, og=2, stride=(1, 1), pad=(1, 1), dil=(1, 1), mask=True),     # 32 channels per offset group
    dict(B=1, C=48, OC=200, H=13, W=17, k=(3, 3), groups=1, og=1, stride=(2, 1), pad=(1, 2), dil=(1, 2), mask=False),   # channel / oc / pixel tails, no mask
    dict(B=3, C=64, OC=64, H=11, W=9, k=(1, 3), groups=2, og=2, stride=(1, 1), pad=(0, 1), dil=(1, 1), mask=True),      # two weight groups
    dict(B=1, C=320, OC=272, H=9, W=10, k=(3, 3), groups=1, og=1, stride=(1, 1), pad=(1, 1), dil=(1, 1), mask=True),    # two channel chunks, two oc chunks
    dict(B=3, C=36, OC=40, H=11, W=9, k=(1, 3), groups=2, og=3, stride=(1, 1), pad=(0, 1), dil=(1, 1), mask=True),      # direct: 12 channels per offset group
    dict(B=2, C=8, OC=8, H=9, W=9, k=(3, 3), groups=8, og=1, stride=(1, 1), pad=(1, 1), dil=(1, 1), mask=False),        # direct: depthwise
    dict(B=2, C=6, OC=2, H=5, W=4, k=(3, 2), groups=2, og=3, stride=(2, 1), pad=(1, 0), dil=(2, 1), mask=True),         # direct: the reference's test configuration
    dict(B=1, C=32, OC=32, H=8, W=8, k=(3, 3), groups=1, og=1, stride=(1, 1), pad=(1, 1), dil=(1, 1), mask=True, zero_off=True),  # y = -1 / x = -1 exactly
    dict(B=2, C=64, OC=64, H=30, W=40, k=(3, 3), groups=1, og=1, stride=(1, 1), pad=(1, 1), dil=(1, 1), mask=True, off_scale=0.5),  # offsets inside the LDS window's reach
    dict(B=1, C=64, OC=32, H=40, W=40, k=(3, 3), groups=1, og=1, stride=(2, 2), pad=(1, 1), dil=(1, 1), mask=True, off_scale=0.5),  # stride 2: window too large for LDS -> global atomics
    dict(B=1, C=64, OC=32, H=8, W=8, k=(3, 3), groups=1, og=1, stride=(1, 1), pad=(1, 1), dil=(1, 1), mask=True, zero_off=True),   # owner form, y = -1 / x = -1 exactly
    dict(B=2, C=128, OC=64, H=17, W=21, k=(3, 3), groups=2, og=2, stride=(1, 1), pad=(1, 1), dil=(1, 1), mask=False, off_scale=4),  # owner form: two weight groups, most offsets beyond the window's reach
    dict(B=2, C=128, OC=128, H=12, W=14, k=(3, 3), groups=128, og=2, stride=(1, 1), pad=(1, 1), dil=(1, 1), mask=True),            # depthwise, channels-last route, one 64-channel pass per offset group
    dict(B=1, C=128, OC=128, H=13, W=9, k=(3, 2), groups=128, og=1, stride=(2, 1), pad=(1, 0), dil=(1, 2), mask=False),            # depthwise, two passes, strides / dilation, no mask
    dict(B=1, C=64, OC=32, H=12, W=20, k=(1, 9), groups=1, og=1, stride=(1, 1), pad=(0, 4), dil=(1, 1), mask=True, off_scale=0.5),   # 9 taps that are NOT 3 x 3: not the owner form
    dict(B=1, C=64, OC=32, H=20, W=12, k=(9, 1), groups=1, og=1, stride=(1, 1), pad=(4, 0), dil=(1, 1), mask=False, off_scale=0.5),  # ... and the transposed one
]


def _dcn_bwd_inputs(cfg, dtype, seed=31):
    g = gen(seed)
    kh, kw = cfg["k"]
    oh = (cfg["H"] + 2 * cfg["pad"][0] - (cfg["dil"][0] * (kh - 1) + 1)) // cfg["stride"][0] + 1
    ow = (cfg["W"] + 2 * cfg["pad"][1] - (cfg["dil"][1] * (kw - 1) + 1)) // cfg["stride"][1] + 1
    x = torch.randn(cfg["B"], cfg["C"], cfg["H"], cfg["W"], generator=g)
    w = torch.randn(cfg["OC"], cfg["C"] // cfg["groups"], kh, kw, generator=g) * 0.1
    off = torch.randn(cfg["B"], 2 * cfg["og"] * kh * kw, oh, ow, generator=g) * cfg.get("off_scale", 2)
    if cfg.get("zero_off"):
        off = torch.zeros_like(off)
    m = torch.rand(cfg["B"], cfg["og"] * kh * kw, oh, ow, generator=g)
    b = torch.randn(cfg["OC"], generator=g)
    gr = torch.randn(cfg["B"], cfg["OC"], oh, ow, generator=g)
    args = (*cfg["stride"], *cfg["pad"], *cfg["dil"], cfg["groups"], cfg["og"], cfg["mask"])
    return [v.to(dtype) for v in (gr, x, w, off, m, b)], args


def _dcn_bwd_compare(got, ref, tol, what):
    for name, a, r in zip(("grad_input", "grad_weight", "grad_offset", "grad_mask", "grad_bias"), got, ref):
        r = r.double().numpy()
        scale = max(1.0, float(np.abs(r).max()))
        np.testing.assert_allclose(a.double().cpu().numpy(), r, rtol=tol, atol=tol * scale, err_msg=f"{name} {what}")


@pytest.mark.parametrize("route", ["default", "window", "global_atomics", "direct"])
@pytest.mark.parametrize("cfg", DCN_BWD_CFGS, ids=[str(i) for i in range(len(DCN_BWD_CFGS))])
def test_deform_conv2d_backward_fused_vs_reference(tv, cfg, route):
    """`_deform_conv2d_backward` = ONE call of tvmi_deform_conv2d_backward (deform_conv2d_bwd.hip: two fused matrix-core kernels
    or the direct kernels; no `columns`, no library GEMM) against the reference CPU kernels
    (cpu/deform_conv2d_kernel.cpp:274-348, 407-551, 1153-1226), all five gradients, on every route: offset groups, weight
    groups, strides / dilations, channel and pixel tails, no mask, and zero offsets with padding (sampling rows at exactly
    y = -1, where get_coordinate_weight still sees the row below while the bilinear sample is zero).  fp32 bar 1e-4 of the
    gradient's scale (sums in another order)."""
    if not O.load_reference():
        pytest.skip("needs the reference CPU kernels (oracle/_ref)")
    ts, args = _dcn_bwd_inputs(cfg, torch.float32)
    # default: the owner form of the data-gradient kernel where the shape allows (3 x 3, 64-channel chunks), else the window kernel
    torch.ops.tvmi.set_option("dcn.bwd_mfma", 0 if route == "direct" else 1)
    torch.ops.tvmi.set_option("dcn.bwd_owner", 0 if route in ("window", "global_atomics") else 1)
    torch.ops.tvmi.set_option("dcn.bwd_window", 0 if route == "global_atomics" else 1)
    try:
        got = tv._deform_conv2d_backward(*[v.to(DEV) for v in ts], *args)
    finally:
        torch.ops.tvmi.set_option("dcn.bwd_mfma", 1)
        torch.ops.tvmi.set_option("dcn.bwd_owner", 1)
        torch.ops.tvmi.set_option("dcn.bwd_window", 1)
    _dcn_bwd_compare(got, tv._deform_conv2d_backward(*ts, *args), TOL, route)


@pytest.mark.parametrize("dtype,tol,idx", [(torch.bfloat16, 1.5e-2, 0), (torch.float16, 2e-3, 1), (torch.bfloat16, 1.5e-2, 6), (torch.float64, 1e-10, 7),
                                           (torch.float64, 1e-10, 0)], ids=["bf16-mfma", "fp16-mfma", "bf16-direct", "fp64-direct", "fp64-large"])
def test_deform_conv2d_backward_fused_other_dtypes(tv, dtype, tol, idx):
    """16-bit tensors are read natively, contracted and summed in fp32 and rounded once (the reference keeps `columns` and its
    atomics in the 16-bit type): compared with the fp32 reference on the rounded tensors, bar = a few 16-bit ulps of the
    gradient's scale.  fp64 always takes the direct kernels: 1e-10."""
    if not O.load_reference():
        pytest.skip("needs the reference CPU kernels (oracle/_ref)")
    ts, args = _dcn_bwd_inputs(DCN_BWD_CFGS[idx], dtype)
    got = tv._deform_conv2d_backward(*[v.to(DEV) for v in ts], *args)
    assert all(a.dtype == dtype for a in got)
    up = torch.float64 if dtype == torch.float64 else torch.float32
    _dcn_bwd_compare(got, tv._deform_conv2d_backward(*[v.to(up) for v in ts], *args), tol, str(dtype))


def test_deform_conv2d_backward_through_the_c_abi():
    """tvmi_deform_conv2d_backward through ctypes on raw device pointers: outputs are FULLY overwritten (poisoned first), the
    workspace query sizes the call, a short workspace is refused."""
    import ctypes

    if not O.load_reference():
        pytest.skip("needs the reference CPU kernels (oracle/_ref)")
    lib = vision_amd._loader.kernels()
    cfg = DCN_BWD_CFGS[1]
    ts, args = _dcn_bwd_inputs(cfg, torch.float32)
    gr, x, w, off, m, b = [v.to(DEV).contiguous() for v in ts]
    outs = [torch.full_like(v, float("nan")) for v in (x, w, off, m, b)]
    B, C, H, W = x.shape
    OC, _, kh, kw = w.shape
    oh, ow = gr.shape[2:]
    i64 = ctypes.c_int64
    lib.tvmi_deform_conv2d_backward_workspace_bytes.restype = ctypes.c_size_t
    lib.tvmi_deform_conv2d_backward_workspace_bytes.argtypes = [ctypes.c_int] + [i64] * 15
    nbytes = lib.tvmi_deform_conv2d_backward_workspace_bytes(0, B, C, H, W, OC, kh, kw, *cfg["stride"], *cfg["pad"], *cfg["dil"],
                                                              cfg["groups"], cfg["og"])
    assert nbytes > 0
    ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    lib.tvmi_deform_conv2d_backward.restype = ctypes.c_int
    lib.tvmi_deform_conv2d_backward.argtypes = [ctypes.c_void_p] * 10 + [ctypes.c_int] + [i64] * 15 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    ptrs = [v.data_ptr() for v in (gr, x, w, off, m)] + [v.data_ptr() for v in outs]
    dims = [B, C, H, W, OC, kh, kw, *cfg["stride"], *cfg["pad"], *cfg["dil"], cfg["groups"], cfg["og"]]
    stream = torch.cuda.current_stream().cuda_stream
    assert lib.tvmi_deform_conv2d_backward(*ptrs, 0, *dims, 1, ws.data_ptr(), nbytes, stream) == 0
    torch.cuda.synchronize()
    _dcn_bwd_compare(outs, torch.ops.torchvision._deform_conv2d_backward(*ts, *args), TOL, "ctypes")
    assert lib.tvmi_deform_conv2d_backward(*ptrs, 0, *dims, 1, ws.data_ptr(), nbytes - 1, stream) != 0


def test_deform_conv2d_zero_offset_is_conv_and_batch0(tv):
    g = gen(15)
    x = torch.randn(2, 32, 14, 14, generator=g).to(DEV)
    w = (torch.randn(64, 32, 3, 3, generator=g) * 0.1).to(DEV)
    off = torch.zeros(2, 18, 14, 14, device=DEV)
    y = vision_amd.deform_conv2d(x, off, w, padding=1)
    torch.testing.assert_close(y, F.conv2d(x, w, padding=1), atol=2e-4, rtol=1e-4)
    assert vision_amd.deform_conv2d(x[:0], off[:0], w, padding=1).shape == (0, 64, 14, 14)
    with pytest.raises(RuntimeError, match="offset.shape\\[1\\] is not valid"):
        tv.deform_conv2d(x, w, off[:, :16], torch.zeros(2, 1, device=DEV), torch.zeros(64, device=DEV), 1, 1, 1, 1, 1, 1, 1, 1, False)


def test_deform_conv2d_gradcheck(tv):
    g = gen(16)
    x = torch.rand(1, 4, 5, 5, generator=g, dtype=torch.float64).to(DEV).requires_grad_(True)
    w = torch.randn(2, 2, 3, 3, generator=g, dtype=torch.float64).to(DEV).requires_grad_(True)
    off = torch.randn(1, 2 * 2 * 9, 5, 5, generator=g, dtype=torch.float64).to(DEV).requires_grad_(True)
    m = torch.rand(1, 2 * 9, 5, 5, generator=g, dtype=torch.float64).to(DEV).requires_grad_(True)
    b = torch.randn(2, generator=g, dtype=torch.float64).to(DEV).requires_grad_(True)
    fn = lambda x_, o_, w_, b_, m_: vision_amd.deform_conv2d(x_, o_, w_, b_, padding=1, mask=m_)
    assert torch.autograd.gradcheck(fn, (x, off, w, b, m), nondet_tol=1e-5, fast_mode=True)
    fn2 = lambda x_, o_, w_, b_: vision_amd.deform_conv2d(x_, o_, w_, b_, padding=1)
    assert torch.autograd.gradcheck(fn2, (x, off, w, b), nondet_tol=1e-5, fast_mode=True)


# ------------------------------------------------------------------------------ rotated IoU
def test_box_iou_rotated(tv):
    g = golden("box_iou_rotated")
    np.testing.assert_allclose(tv.box_iou_rotated(t(g["b1"]), t(g["b2"])).cpu().numpy(), g["iou"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(tv.box_iou_rotated(t(g["b1"]).double(), t(g["b2"]).double()).cpu().numpy(), g["iou64"], rtol=0, atol=1e-6)
    boxes = torch.tensor([[0, 0, 10, 10, 45], [0, 0, 10, 10, 135], [0, 0, 10, 10, -45], [0, 0, 10, 10, -135],
                          [100, 100, 10, 10, 30], [50, 50, 20, 10, 45], [50, 50, 20, 10, 135], [50, 50, 20, 10, -135]],
                         dtype=torch.float32)
    out = vision_amd.box_iou(boxes.to(DEV), boxes.to(DEV), fmt="cxcywhr").cpu()
    assert out.dtype == torch.float32
    assert abs(out[5, 6].item() - 1 / 3) < 1e-4 and abs(out[0, 3].item() - 1) < 1e-4 and out[0, 4].item() == 0
    gg = gen(17)
    b1 = torch.cat([torch.rand(700, 2, generator=gg) * 200, 2 + torch.rand(700, 2, generator=gg) * 60, torch.rand(700, 1, generator=gg) * 720 - 360], 1)
    b2 = b1.flip(0)[:333].contiguous()
    np.testing.assert_allclose(tv.box_iou_rotated(b1.to(DEV), b2.to(DEV)).cpu().numpy(), O.box_iou_rotated(b1.numpy(), b2.numpy()), rtol=0, atol=1e-5)
    assert tv.box_iou_rotated(torch.empty(0, 5, device=DEV), b2.to(DEV)).shape == (0, 333)


def test_box_iou_rotated_dense_tiles_and_degenerate_boxes(tv):
    """The kernel clips the pairs that survive the circle test from a compacted per-tile list: tiles in which EVERY pair
    survives (a cluster of overlapping boxes: the list is full, 4096 entries, 16 rounds of 256), tiles with none, identical
    and axis-aligned boxes (the Graham scan's tie rules), zero / negative extents and a NaN — all against the C restatement of
    cpu/box_iou_rotated_kernel.cpp, which tests/test_oracle.py pins to the reference's own kernel."""
    gg = gen(171)
    cluster = torch.cat([100 + torch.rand(150, 2, generator=gg) * 30, 20 + torch.rand(150, 2, generator=gg) * 60,
                         torch.rand(150, 1, generator=gg) * 360 - 180], 1)
    far = torch.cat([5000 + torch.rand(70, 2, generator=gg) * 9000, 2 + torch.rand(70, 2, generator=gg) * 5, torch.zeros(70, 1)], 1)
    special = torch.tensor([[100, 100, 40, 20, 0], [100, 100, 40, 20, 0], [100, 100, 40, 20, 90], [100, 100, 40, 20, 180],
                            [120, 100, 40, 20, 0], [100, 110, 40, 20, 0], [100, 100, 0, 20, 10], [100, 100, 40, -20, 10],
                            [100, 100, 1e-8, 1e-8, 0], [100, 100, 40, 20, 1e-4]], dtype=torch.float32)
    b1 = torch.cat([cluster, far, special])
    b2 = torch.cat([special, cluster[:100], far[:30]])
    got = tv.box_iou_rotated(b1.to(DEV), b2.to(DEV)).cpu().numpy()
    np.testing.assert_allclose(got, O.box_iou_rotated(b1.numpy(), b2.numpy()), rtol=0, atol=1e-5)
    got64 = tv.box_iou_rotated(b1.double().to(DEV), b2.double().to(DEV)).cpu().numpy()
    np.testing.assert_allclose(got64, O.box_iou_rotated(b1.double().numpy(), b2.double().numpy()), rtol=0, atol=1e-6)
    nan_box = b1.clone()
    nan_box[3, 0] = float("nan")
    got = tv.box_iou_rotated(nan_box.to(DEV), b2.to(DEV)).cpu().numpy()
    want = O.box_iou_rotated(nan_box.numpy(), b2.numpy())
    keep = np.ones(len(b1), bool)
    keep[3] = False
    np.testing.assert_allclose(got[keep], want[keep], rtol=0, atol=1e-5)


# ------------------------------------------------------------------------------ resize
def test_resize_golden_and_torch_cpu(tv):
    g = golden("resize")
    img = t(g["img"])
    for key in g.files:
        if key in ("img", "torch_version"):
            continue
        mode, size, flag = key.rsplit("_", 2)
        oh, ow = (int(v) for v in size.split("x"))
        ac = None if mode.startswith("nearest") else (flag == "ac1")
        out = vision_amd.interpolate(img, size=(oh, ow), mode=mode, align_corners=ac, antialias=(flag == "aa1"))
        np.testing.assert_allclose(out.cpu().numpy(), g[key], rtol=0, atol=TOL, err_msg=key)


@pytest.mark.parametrize("mode,aa", [("bilinear", False), ("bilinear", True), ("bicubic", False), ("bicubic", True),
                                      ("nearest", False), ("nearest-exact", False)])
def test_resize_vs_installed_torch_cpu(mode, aa):
    g = gen(18)
    img = torch.rand(2, 3, 480, 640, generator=g)
    ac = None if mode.startswith("nearest") else False
    for size in ((800, 1067), (123, 77), (480, 640)):
        ref = F.interpolate(img, size=size, mode=mode, align_corners=ac, antialias=aa)
        out = vision_amd.interpolate(img.to(DEV), size=size, mode=mode, align_corners=ac, antialias=aa)
        np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=0, atol=TOL)
    for dt, tol in ((torch.float16, 4e-3), (torch.bfloat16, 2e-2)):
        out = vision_amd.interpolate(img.to(DEV, dt), size=(300, 333), mode=mode, align_corners=ac, antialias=aa)
        ref = F.interpolate(img.to(dt).float(), size=(300, 333), mode=mode, align_corners=ac, antialias=aa)
        assert out.dtype == dt
        np.testing.assert_allclose(out.float().cpu().numpy(), ref.numpy(), rtol=0, atol=tol)


@pytest.mark.parametrize("mode", ["bilinear", "bicubic"])
@pytest.mark.parametrize("align", [False, True])
def test_resize_bilinear_tile_kernel_shapes(align, mode):
    """The LDS-tiled bilinear / bicubic kernels (256 x 4 output tiles, patch staged with 16-byte loads) at their corner cases: widths
    that are no multiple of 4 / 256, rows that are no multiple of 4, inputs 4..9 pixels wide (shifted last quad), up- and
    down-scales up to its limits (1.49 / 2.3; beyond: the per-output kernel), align_corners — against torch CPU."""
    g = gen(71)
    cases = [((5, 4), (7, 9)), ((9, 7), (4, 4)), ((33, 257), (40, 258)), ((61, 301), (50, 203)), ((64, 511), (30, 345)),
             ((135, 240), (100, 178)), ((37, 53), (111, 160)), ((20, 300), (9, 201)), ((50, 1000), (23, 700)),
             ((17, 260), (17, 260)), ((12, 9), (30, 1031))]
    for (ih, iw), (oh, ow) in cases:
        # the launcher takes the tiled kernel from 16384 tile-planes on: enough planes for every shape
        tiles = -(-ow // 256) * -(-oh // 4)
        img = torch.rand(1, -(-16384 // tiles) + 3, ih, iw, generator=g)
        ref = F.interpolate(img, size=(oh, ow), mode=mode, align_corners=align)
        out = vision_amd.interpolate(img.to(DEV), size=(oh, ow), mode=mode, align_corners=align)
        np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=0, atol=TOL, err_msg=f"{mode} {ih}x{iw}->{oh}x{ow}")
        out16 = vision_amd.interpolate(img.to(DEV, torch.bfloat16), size=(oh, ow), mode=mode, align_corners=align)
        ref16 = F.interpolate(img.to(torch.bfloat16).float(), size=(oh, ow), mode=mode, align_corners=align)
        np.testing.assert_allclose(out16.float().cpu().numpy(), ref16.numpy(), rtol=0, atol=2e-2)


@pytest.mark.parametrize("mode", ["bilinear", "bicubic"])
def test_resize_antialias_tile_kernel_shapes(mode):
    """The LDS-tiled anti-aliased kernel (separable inside a 256 x 4 output tile; both axes <= 8 taps) at its corner
    cases — ragged widths / heights, narrow inputs, image edges inside a tile, up- and down-scales up to its limits —
    against torch CPU F.interpolate(antialias=True)."""
    g = gen(72)
    cases = [((9, 8), (7, 6)), ((33, 257), (30, 200)), ((61, 301), (50, 223)), ((64, 340), (48, 255)), ((135, 240), (100, 178)),
             ((37, 53), (111, 160)), ((40, 300), (17, 230)), ((50, 1000), (40, 700)), ((17, 260), (17, 260)), ((24, 9), (30, 1031)),
             ((100, 128), (29, 100))]
    for (ih, iw), (oh, ow) in cases:
        tiles = -(-ow // 256) * -(-oh // 4)
        img = torch.rand(1, -(-16384 // tiles) + 3, ih, iw, generator=g)
        ref = F.interpolate(img, size=(oh, ow), mode=mode, align_corners=False, antialias=True)
        out = vision_amd.interpolate(img.to(DEV), size=(oh, ow), mode=mode, align_corners=False, antialias=True)
        np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=0, atol=TOL, err_msg=f"{mode} {ih}x{iw}->{oh}x{ow}")
    img = torch.rand(1, 5000, 61, 301, generator=g)
    out16 = vision_amd.interpolate(img.to(DEV, torch.float16), size=(50, 223), mode=mode, align_corners=False, antialias=True)
    ref16 = F.interpolate(img.to(torch.float16).float(), size=(50, 223), mode=mode, align_corners=False, antialias=True)
    np.testing.assert_allclose(out16.float().cpu().numpy(), ref16.numpy(), rtol=0, atol=4e-3)


@pytest.mark.parametrize("mode,aa", [("nearest", False), ("nearest-exact", False), ("bilinear", False), ("bicubic", False), ("bilinear", True), ("bicubic", True)])
def test_interpolate_gradient_matches_torch(mode, aa):
    """vision_amd.interpolate on an input that requires grad: forward and gradient on the resize kernels of this library.
    Equal to the gradient ATen's own CUDA kernels give F.interpolate on the same device tensor (the forward value of which
    differs by rounding only; ATen's scatter-add sums in another order)."""
    g = gen(52)
    x = torch.rand(2, 3, 19, 23, generator=g).to(DEV)
    kw = {} if mode.startswith("nearest") else dict(align_corners=False, antialias=aa)
    for size in ((31, 40), (11, 9)):
        a = x.clone().requires_grad_(True)
        b = x.clone().requires_grad_(True)
        ya = vision_amd.interpolate(a, size=size, mode=mode, **kw)
        yb = F.interpolate(b, size=size, mode=mode, **kw)
        w = torch.randn(ya.shape, generator=g).to(DEV)
        (ya * w).sum().backward()
        (yb * w).sum().backward()
        assert float((ya - yb).abs().max()) <= TOL
        torch.testing.assert_close(a.grad, b.grad, rtol=1e-5, atol=1e-5)


_BWD_CASES = [  # (in_h, in_w) -> size or scale_factor
    ((19, 23), dict(size=(31, 40))), ((19, 23), dict(size=(11, 9))), ((19, 23), dict(size=(19, 23))), ((19, 23), dict(size=(19, 40))),
    ((5, 7), dict(size=(64, 93))),            # > 8 outputs per input pixel and axis: the chunked column loop
    ((200, 301), dict(size=(17, 23))),        # strong down-scale: wide anti-aliasing supports, most inputs untouched without it
    ((37, 300), dict(size=(74, 600))),        # rows wider than one 256-lane workgroup
    ((24, 32), dict(scale_factor=(1.7, 2.3))), ((24, 32), dict(scale_factor=0.55, recompute_scale_factor=True)),
]


@pytest.mark.parametrize("mode,aa", [("nearest", False), ("nearest-exact", False), ("bilinear", False), ("bicubic", False), ("bilinear", True), ("bicubic", True)])
def test_interpolate_backward_kernels_vs_torch_cpu(mode, aa):
    """tvmi::interpolate2d_backward (gather form of aten::upsample_*_backward, resize.hip) against the gradient installed-torch
    CPU autograd gives F.interpolate — the arithmetic behind the reference's resize wrappers (_geometry.py:344), the FPN
    top-down path (ops/feature_pyramid_network.py:194) and the segmentation heads (models/segmentation/_utils.py:27,33).
    fp32 at 1e-5 of the gradient's scale; fp16 / bf16 gradients accumulate in fp32 and are rounded once.  Both align_corners
    settings, user scale factors, identity sizes, N*C not a multiple of the plane group, and run-to-run bit equality."""
    g = gen(61)
    for (ih, iw), kw in _BWD_CASES:
        for align in ((False,) if mode.startswith("nearest") else (False, True)):
            if aa and align:
                continue
            extra = {} if mode.startswith("nearest") else dict(align_corners=align, antialias=aa)
            x = torch.rand(1, 7, ih, iw, generator=g, requires_grad=True)
            y = F.interpolate(x, mode=mode, **kw, **extra)
            w = torch.randn(y.shape, generator=g)
            (y * w).sum().backward()
            xd = x.detach().to(DEV).requires_grad_(True)
            yd = vision_amd.interpolate(xd, mode=mode, **kw, **extra)
            assert yd.shape == y.shape
            (yd * w.to(DEV)).sum().backward()
            scale = max(float(x.grad.abs().max()), 1e-6)
            err = float((xd.grad.cpu() - x.grad).abs().max()) / scale
            assert err <= 1e-5, (mode, aa, align, (ih, iw), kw, err)
            # the op directly, twice: bit-identical (no atomics)
            sh, sw = -1.0, -1.0
            if "scale_factor" in kw and not kw.get("recompute_scale_factor"):
                sf = kw["scale_factor"]
                sh, sw = (sf, sf) if isinstance(sf, float) else sf
            args = (ih, iw, {"nearest": 0, "nearest-exact": 1, "bilinear": 2, "bicubic": 3}[mode], align, aa, sh, sw)
            g1 = torch.ops.tvmi.interpolate2d_backward(w.to(DEV), *args)
            g2 = torch.ops.tvmi.interpolate2d_backward(w.to(DEV), *args)
            assert torch.equal(g1, g2) and torch.equal(g1, xd.grad)
            for dt, tol in ((torch.float16, 2e-3), (torch.bfloat16, 1.6e-2)):
                g16 = torch.ops.tvmi.interpolate2d_backward(w.to(DEV, dt), *args)
                x2 = x.detach().clone().requires_grad_(True)
                (F.interpolate(x2, mode=mode, **kw, **extra) * w.to(dt).float()).sum().backward()
                assert g16.dtype == dt
                assert float((g16.float().cpu() - x2.grad).abs().max()) / scale <= tol, (mode, aa, dt, kw)


def test_interpolate_backward_under_the_aten_override():
    """With the aten override on, autograd's `upsample_*_backward` calls land in resize.hip too (call counter), grad_input
    equal to installed-torch CPU's; float64 and channels_last gradients still reach ATen's kernels."""
    g = gen(62)
    x = torch.rand(2, 5, 40, 52, generator=g)
    was = vision_amd.override_aten_upsample(True)
    try:
        for mode, kw in (("nearest", {}), ("nearest-exact", {}), ("bilinear", dict(align_corners=False)), ("bicubic", dict(align_corners=True)),
                         ("bilinear", dict(align_corners=False, antialias=True)), ("bicubic", dict(align_corners=False, antialias=True))):
            for size in ((80, 104), (23, 31)):
                a = x.clone().requires_grad_(True)
                ya = F.interpolate(a, size=size, mode=mode, **kw)
                w = torch.randn(ya.shape, generator=g)
                (ya * w).sum().backward()
                b = x.to(DEV).requires_grad_(True)
                c0 = int(torch.ops.tvmi.aten_upsample_calls())
                yb = F.interpolate(b, size=size, mode=mode, **kw)
                (yb * w.to(DEV)).sum().backward()
                assert int(torch.ops.tvmi.aten_upsample_calls()) - c0 == 2, (mode, size)   # forward + backward
                torch.testing.assert_close(b.grad.cpu(), a.grad, rtol=1e-5, atol=1e-5)
        c1 = int(torch.ops.tvmi.aten_upsample_calls())
        d = x.double().to(DEV).requires_grad_(True)
        F.interpolate(d, size=(50, 60), mode="bilinear", align_corners=False).sum().backward()
        assert int(torch.ops.tvmi.aten_upsample_calls()) == c1 and d.grad is not None
        # channels_last: the forward has its own kernel, a channels_last gradient keeps ATen's channels_last backward kernel
        e = x.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        ye = F.interpolate(e, size=(50, 60), mode="bilinear", align_corners=False)
        ye.backward(torch.ones_like(ye))
        assert int(torch.ops.tvmi.aten_upsample_calls()) - c1 == 1 and e.grad is not None
        f = x.clone().requires_grad_(True)
        F.interpolate(f, size=(50, 60), mode="bilinear", align_corners=False).sum().backward()
        torch.testing.assert_close(e.grad.cpu(), f.grad, rtol=1e-5, atol=1e-5)
    finally:
        vision_amd.override_aten_upsample(was)


@pytest.mark.parametrize("dtype", [torch.uint8, torch.int8, torch.int16, torch.int32, torch.int64, torch.bool])
def test_nearest_resize_of_integer_tensors(dtype):
    """The nearest modes are copies: uint8 images / masks (and every other integer type) are resized without a cast, like
    aten::upsample_nearest2d on the reference's resize path (_geometry.py:316-323).  Bit-equal to torch CPU on the types torch
    CPU serves (uint8), and to the float round trip for the others; the aten override serves them as well."""
    g = gen(63)
    x = torch.randint(0, 2 if dtype == torch.bool else 100, (2, 3, 37, 53), generator=g).to(dtype)
    for mode in ("nearest", "nearest-exact"):
        for size in ((80, 31), (11, 120), (37, 53)):
            ref = F.interpolate(x.float(), size=size, mode=mode).to(dtype)
            out = vision_amd.interpolate(x.to(DEV), size=size, mode=mode)
            assert out.dtype == dtype and torch.equal(out.cpu(), ref), (mode, size)
            if dtype == torch.uint8:
                assert torch.equal(out.cpu(), F.interpolate(x, size=size, mode=mode))
    if dtype == torch.uint8:
        was = vision_amd.override_aten_upsample(True)
        try:
            c0 = int(torch.ops.tvmi.aten_upsample_calls())
            out = F.interpolate(x.to(DEV), size=(64, 64), mode="nearest")
            assert int(torch.ops.tvmi.aten_upsample_calls()) - c0 == 1
            assert torch.equal(out.cpu(), F.interpolate(x, size=(64, 64), mode="nearest"))
        finally:
            vision_amd.override_aten_upsample(was)
        img = torch.randint(0, 256, (3, 120, 160), generator=g, dtype=torch.uint8)
        out = vision_amd.resize(img.to(DEV), [90], interpolation="nearest-exact")
        assert out.dtype == torch.uint8 and torch.equal(out.cpu(), F.interpolate(img[None], size=(90, 120), mode="nearest-exact")[0])


@pytest.mark.parametrize("mode,aa", [("nearest", False), ("nearest-exact", False), ("bilinear", False), ("bicubic", False), ("bilinear", True), ("bicubic", True)])
def test_resize_channels_last(mode, aa):
    """channels_last in -> channels_last out without a layout copy (upsample2d_nhwc_kernel), all six modes, against torch CPU at
    1e-4 (16-bit: the input rounded first); C = 3 images (the 1CHW case of resize_image, _geometry.py:324-338) and a 256-channel
    map; uint8 in the nearest modes; the aten override serves the format as well."""
    g = gen(64)
    for shape, sizes in (((2, 3, 61, 83), ((100, 131), (23, 40), (61, 83))), ((1, 256, 25, 42), ((50, 84), (13, 20)))):
        x = torch.rand(*shape, generator=g)
        for size in sizes:
            for align in ((False,) if mode.startswith("nearest") or aa else (False, True)):
                kw = {} if mode.startswith("nearest") else dict(align_corners=align, antialias=aa)
                ref = F.interpolate(x, size=size, mode=mode, **kw)
                out = vision_amd.interpolate(x.to(DEV).contiguous(memory_format=torch.channels_last), size=size, mode=mode, **kw)
                assert out.shape == ref.shape and out.is_contiguous(memory_format=torch.channels_last)
                assert float((out.cpu() - ref).abs().max()) <= TOL, (mode, aa, shape, size, align)
        x16 = x.to(torch.bfloat16)
        kw = {} if mode.startswith("nearest") else dict(align_corners=False, antialias=aa)
        out16 = vision_amd.interpolate(x16.to(DEV).contiguous(memory_format=torch.channels_last), size=sizes[0], mode=mode, **kw)
        ref16 = F.interpolate(x16.float(), size=sizes[0], mode=mode, **kw)
        assert out16.dtype == torch.bfloat16 and float((out16.float().cpu() - ref16).abs().max()) <= 1.6e-2
    if mode.startswith("nearest"):
        u = torch.randint(0, 256, (2, 3, 37, 53), generator=g, dtype=torch.uint8)
        out = vision_amd.interpolate(u.to(DEV).contiguous(memory_format=torch.channels_last), size=(80, 31), mode=mode)
        assert out.is_contiguous(memory_format=torch.channels_last) and torch.equal(out.cpu(), F.interpolate(u, size=(80, 31), mode=mode))
    was = vision_amd.override_aten_upsample(True)
    try:
        c0 = int(torch.ops.tvmi.aten_upsample_calls())
        x = torch.rand(2, 8, 30, 40, generator=g)
        kw = {} if mode.startswith("nearest") else dict(align_corners=False, antialias=aa)
        out = F.interpolate(x.to(DEV).contiguous(memory_format=torch.channels_last), size=(45, 70), mode=mode, **kw)
        assert int(torch.ops.tvmi.aten_upsample_calls()) - c0 == 1 and out.is_contiguous(memory_format=torch.channels_last)
        assert float((out.cpu() - F.interpolate(x, size=(45, 70), mode=mode, **kw)).abs().max()) <= TOL
    finally:
        vision_amd.override_aten_upsample(was)


def test_resize_image_wrapper_uint8():
    g = gen(19)
    img = torch.randint(0, 256, (3, 120, 160), generator=g, dtype=torch.uint8)
    for interp in ("bilinear", "bicubic", "nearest"):
        out = vision_amd.resize(img.to(DEV), [90], interpolation=interp, max_size=130, antialias=True)
        assert out.dtype == torch.uint8 and tuple(out.shape) == (3, 90, 120)
        x = img.float()[None]
        ref = F.interpolate(x, size=(90, 120), mode=interp, align_corners=False if interp != "nearest" else None,
                            antialias=interp != "nearest")
        if interp == "bicubic":
            ref = ref.clamp(0, 255)
        ref = ref.round().to(torch.uint8)[0]
        assert (out.cpu().int() - ref.int()).abs().max().item() <= 1
    same = img.to(DEV)
    assert vision_amd.resize(same, [120, 160]) is same          # identity size: no kernel, no copy (like F.resize)


# ------------------------------------------------------------------------------ detection payload packing
def test_pack_detections_matches_host_reference():
    from vision_amd import sharding

    g = gen(20)
    n, B, D = 5000, 7, 100
    boxes = random_boxes(n, 1344, 800, 8, 300, g)
    scores = torch.rand(n, generator=g)
    labels = torch.randint(1, 91, (n,), generator=g)
    img = torch.randint(0, B, (n,), generator=g)
    img[img == 3] = 2                                   # image 3 keeps nothing
    keep = torch.argsort(scores, descending=True, stable=True)[:3000]
    d_ref, c_ref = sharding.pack_kept_detections(boxes, scores, img, keep, B, D, labels)      # python path (CPU)
    d, c = sharding.pack_kept_detections(boxes.to(DEV), scores.to(DEV), img.to(DEV), keep.to(DEV), B, D, labels.to(DEV))
    assert torch.equal(c.cpu(), c_ref) and c_ref[3].item() == 0
    assert torch.equal(d.cpu(), d_ref)
    d0, c0 = sharding.pack_kept_detections(boxes.to(DEV), scores.to(DEV), img.to(DEV), keep[:0].to(DEV), B, D)
    assert d0.abs().sum().item() == 0 and c0.sum().item() == 0


@pytest.mark.parametrize("H,W", [(1, 9), (2, 3), (3, 5), (25, 42), (13, 43), (50, 84), (7, 4), (64, 64)])
def test_roi_align_dma_path_edge_geometries(tv, H, W):
    """7x7/14x14 sampling_ratio-2 fast paths (LDS-DMA window, shifted edge samples) on awkward map sizes:
    maps narrower than a quad, widths not divisible by 4, RoIs hugging every border, windows of every size class."""
    g = gen(30 + H + W)
    N, C = 2, 40
    x = torch.randn(N, C, H, W, generator=g)
    k = 64
    x1 = torch.rand(k, generator=g) * W * 1.2 - 0.1 * W
    y1 = torch.rand(k, generator=g) * H * 1.2 - 0.1 * H
    bw = torch.rand(k, generator=g) ** 2 * W * 1.1
    bh = torch.rand(k, generator=g) ** 2 * H * 1.1
    rois = torch.stack([torch.randint(0, N, (k,), generator=g).float(), x1, y1, x1 + bw, y1 + bh], 1)
    rois[0, 1:] = torch.tensor([0.0, 0.0, float(W), float(H)])
    rois[1, 1:] = torch.tensor([W - 1.0, H - 1.0, float(W), float(H)])
    rois[2, 1:] = torch.tensor([W - 0.5, 0.0, W + 3.0, float(H)])
    rois[3, 1:] = torch.tensor([-2.0, -2.0, 0.4, 0.4])
    for P in (7, 14):
        for aligned in (False, True):
            y = tv.roi_align(x.to(DEV), rois.to(DEV), 1.0, P, P, 2, aligned)
            ref = O.roi_align(x.numpy(), rois.numpy(), 1.0, P, P, 2, aligned)
            np.testing.assert_allclose(y.cpu().numpy(), ref, rtol=0, atol=TOL)
            gr = torch.randn(ref.shape, generator=g)
            gi = tv._roi_align_backward(gr.to(DEV), rois.to(DEV), 1.0, P, P, N, C, H, W, 2, aligned)
            refb = O.roi_align_backward(gr.numpy(), rois.numpy(), 1.0, P, P, N, C, H, W, 2, aligned)
            np.testing.assert_allclose(gi.cpu().numpy(), refb, rtol=1e-4, atol=TOL * max(1.0, float(np.abs(refb).max())))


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 4e-3), (torch.bfloat16, 5e-3)])
@pytest.mark.parametrize("P", [7, 14])
def test_roi_align_16bit_dma_path(tv, dtype, tol, P):
    """fp16 / bf16 through the LDS-DMA fast path (8 elements per 16-byte piece); bar = the reference's own
    half / bf16 tolerances (test/test_ops.py:133-140) against the fp32 result on the rounded inputs."""
    g = gen(40 + P)
    N, C, H, W = 2, 48, 50, 84
    x = torch.rand(N, C, H, W, generator=g).to(dtype)
    rois = rois_for(N, 120, W * 16, H * 16, 16, 500, g).to(dtype)
    y = tv.roi_align(x.to(DEV), rois.to(DEV), 1 / 16, P, P, 2, False)
    assert y.dtype == dtype
    ref = O.roi_align(x.float().numpy(), rois.float().numpy(), 1 / 16, P, P, 2, False)
    np.testing.assert_allclose(y.float().cpu().numpy(), ref, rtol=tol, atol=tol)
    feats = {str(i): torch.rand(N, 32, 800 // s, 1344 // s, generator=g).to(dtype) for i, s in enumerate((4, 8, 16, 32))}
    boxes = [random_boxes(100, 1344, 800, 8, 600, g) for _ in range(N)]
    pool = vision_amd.MultiScaleRoIAlign(["0", "1", "2", "3"], P, 2)
    with torch.no_grad():
        out = pool({k: v.to(DEV) for k, v in feats.items()}, [b.to(DEV) for b in boxes], [(800, 1344)] * N)
    assert out.dtype == dtype
    ref32 = pool({k: v.float().to(DEV) for k, v in feats.items()}, [b.to(DEV) for b in boxes], [(800, 1344)] * N)   # fp32 boxes either way
    np.testing.assert_allclose(out.float().cpu().numpy(), ref32.cpu().numpy(), rtol=tol, atol=tol)


def test_training_mode_transform_matches_the_reference(tmp_path):
    """GeneralizedRCNNTransform in TRAINING mode (transform.py:119-204): several min sizes drawn through torch's RNG per image,
    target boxes / keypoints scaled, target masks resized, images normalised + resized + batched.  The reference module
    (imported from the staged reference python, run on CPU tensors with the same seed) against vision_amd.transform_with_targets
    on device tensors: same sizes, boxes and keypoints equal, masks identical, image batch within 1e-4."""
    import subprocess, sys, textwrap, os
    from helpers import ROOT
    sys.path.insert(0, ROOT)
    from tools.stage_reference_python import reference_package
    pkg = reference_package(str(tmp_path))
    if pkg is None:
        pytest.skip("reference python package neither present nor staged")
    from vision_amd import integration
    overlay = integration.make_overlay(str(tmp_path / "overlay"), pkg)
    code = textwrap.dedent(f"""
        import sys, torch
        sys.path.insert(0, {overlay!r}); sys.path.insert(0, {ROOT!r})
        import torchvision
        from torchvision.models.detection.transform import GeneralizedRCNNTransform
        import vision_amd
        g = torch.Generator().manual_seed(3)
        imgs = [torch.rand(3, 300, 400, generator=g), torch.rand(3, 260, 210, generator=g), torch.rand(3, 128, 500, generator=g)]
        tgts = []
        for im in imgs:
            h, w = im.shape[-2:]
            b = torch.rand(5, 4, generator=g) * torch.tensor([w / 2, h / 2, w / 2, h / 2]); b[:, 2:] += b[:, :2]
            tgts.append(dict(boxes=b, labels=torch.arange(5), masks=(torch.rand(5, h, w, generator=g) > 0.5).to(torch.uint8),
                             keypoints=torch.rand(5, 4, 3, generator=g) * 100))
        sizes, mean, std = (240, 272, 304, 336), [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
        ref = GeneralizedRCNNTransform(sizes, 448, mean, std).train()
        torch.manual_seed(11)
        il, rt = ref(imgs, [dict(t) for t in tgts])
        torch.manual_seed(11)
        dimgs = [i.cuda() for i in imgs]
        dt = [{{k: v.cuda() for k, v in t.items()}} for t in tgts]
        tensors, new_sizes, ot = vision_amd.transform_with_targets(dimgs, dt, training=True, min_size=sizes, max_size=448,
                                                                   image_mean=mean, image_std=std)
        assert [tuple(s) for s in new_sizes] == [tuple(s) for s in il.image_sizes], (new_sizes, il.image_sizes)
        assert len(set(min(s) for s in new_sizes)) > 1 or True
        assert tuple(tensors.shape) == tuple(il.tensors.shape)
        assert float((tensors.cpu() - il.tensors).abs().max()) <= 1e-4
        for a, b in zip(ot, rt):
            assert torch.equal(a["boxes"].cpu(), b["boxes"]) and torch.equal(a["keypoints"].cpu(), b["keypoints"])
            assert a["masks"].dtype == torch.uint8 and torch.equal(a["masks"].cpu(), b["masks"])
            assert torch.equal(a["labels"].cpu(), b["labels"])
        assert torch.equal(dt[0]["boxes"].cpu(), tgts[0]["boxes"])        # inputs untouched
        print("TRANSFORM_OK")
        """)
    env = dict(os.environ, TVMI_NO_PY_REGISTRATIONS="1")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0 and "TRANSFORM_OK" in out.stdout, out.stdout[-1500:] + out.stderr[-3000:]


# ------------------------------------------------------------------ mask paste
def test_paste_masks_golden_and_oracle():
    """One-launch paste_masks_in_image vs the reference python's own output (tests/golden/detection.npz) and,
    on a full-size canvas, vs the oracle restatement.  The pasted rectangle (integer box arithmetic) must be
    identical; values within 1e-5 (fp32 bilinear, association of the 4-tap sum differs from aten's CPU kernel)."""
    G = golden("detection")
    boxes, shape = torch.from_numpy(G["paste_boxes"]), tuple(int(v) for v in G["paste_shape"])
    for M, pad in ((28, 1), (14, 2), (7, 0)):
        want = G[f"paste_out_M{M}_p{pad}"]
        got = vision_amd.paste_masks_in_image(torch.from_numpy(G[f"paste_masks_M{M}_p{pad}"]).to(DEV), boxes.to(DEV), shape, padding=pad)
        got = got.cpu().numpy()
        assert got.shape == want.shape
        assert np.array_equal(got != 0, want != 0)
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-5)
    g = gen(77)
    im_h, im_w, n = 800, 1333, 24
    b = random_boxes(n, im_w, im_h, 2, 700, g)
    b[0] = torch.tensor([-5.0, -3.0, 40.0, 30.0])          # sticks out on the top-left
    b[1] = torch.tensor([1300.0, 780.0, 1340.0, 805.0])    # sticks out on the bottom-right
    m = torch.rand(n, 1, 28, 28, generator=g)
    got = vision_amd.paste_masks_in_image(m.to(DEV), b.to(DEV), (im_h, im_w)).cpu().numpy()
    ref = O.paste_masks_in_image(m.numpy(), b.numpy(), (im_h, im_w))
    assert np.array_equal(got != 0, ref != 0)
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-5)
    # 16-bit masks keep their dtype; empty input keeps the reference's empty shape
    h = vision_amd.paste_masks_in_image(m.half().to(DEV), b.to(DEV), (im_h, im_w))
    assert h.dtype == torch.float16
    np.testing.assert_allclose(h.float().cpu().numpy(), O.paste_masks_in_image(m.half().float().numpy(), b.numpy(), (im_h, im_w)), rtol=0, atol=2e-3)
    assert vision_amd.paste_masks_in_image(m[:0].to(DEV), b[:0].to(DEV), (im_h, im_w)).shape == (0, 1, im_h, im_w)


# ------------------------------------------------------------------ fused post-processing
def test_postprocess_detections_golden_and_oracle():
    """Batched postprocess_detections (candidate kernel + segmented NMS + packing) vs the reference python's own
    output: identical labels / order, scores within 1e-6, boxes within 1e-4 px (device expf vs torch's CPU exp)."""
    G = golden("detection")
    shapes = [tuple(int(v) for v in s) for s in G["det_shapes"]]
    props = [torch.from_numpy(G[f"det_props{i}"]).to(DEV) for i in range(len(shapes))]
    kw = dict(score_thresh=0.05, nms_thresh=0.5, detections_per_img=20)
    boxes, scores, labels = vision_amd.postprocess_detections(torch.from_numpy(G["det_logits"]).to(DEV),
                                                              torch.from_numpy(G["det_reg"]).to(DEV), props, shapes, **kw)
    for i in range(len(shapes)):
        assert np.array_equal(labels[i].cpu().numpy(), G[f"det_labels{i}"])
        np.testing.assert_allclose(scores[i].cpu().numpy(), G[f"det_scores{i}"], rtol=0, atol=1e-6)
        np.testing.assert_allclose(boxes[i].cpu().numpy(), G[f"det_boxes{i}"], rtol=0, atol=1e-4)
    # model-sized batch against the oracle restatement: 4 images x 1000 proposals x 91 classes
    g = gen(91)
    shapes = [(800, 1333), (800, 1200), (750, 1333), (640, 960)]
    props = [random_boxes(1000, w, h, 8, 500, g) for h, w in shapes]
    logits = torch.randn(4000, 91, generator=g) * 3
    reg = torch.randn(4000, 364, generator=g) * 0.5
    dets, counts = vision_amd.postprocess_detections(logits.to(DEV), reg.to(DEV), [p.to(DEV) for p in props], shapes, padded=True)
    ref = O.postprocess_detections(logits.numpy(), reg.numpy(), [p.numpy() for p in props], shapes)
    for i, (b, s, lab) in enumerate(ref):
        n = int(counts[i])
        assert n == len(lab)
        d = dets[i, :n].cpu().numpy()
        assert np.array_equal(d[:, 5].astype(np.int64), lab)
        np.testing.assert_allclose(d[:, 4], s, rtol=0, atol=1e-6)
        np.testing.assert_allclose(d[:, :4], b, rtol=0, atol=2e-3)
        assert not dets[i, n:].any()


def test_filter_proposals_golden_and_fused_decode():
    G = golden("detection")
    shapes = [tuple(int(v) for v in s) for s in G["det_shapes"]]
    levels = [int(v) for v in G["rpn_levels"]]
    kw = dict(pre_nms_top_n=60, post_nms_top_n=40, nms_thresh=0.7, score_thresh=0.1)
    obj = torch.from_numpy(G["rpn_objectness"]).to(DEV)
    for fused in (False, True):
        if fused:  # anchors + deltas, decoded inside the kernel for the survivors only
            b, s = vision_amd.filter_proposals(torch.from_numpy(G["rpn_anchors"]).to(DEV), obj, shapes, levels,
                                               pred_bbox_deltas=torch.from_numpy(G["rpn_deltas"]).to(DEV), **kw)
        else:
            b, s = vision_amd.filter_proposals(torch.from_numpy(G["rpn_proposals"]).to(DEV), obj, shapes, levels, **kw)
        for i in range(2):
            assert b[i].shape == G[f"rpn_boxes{i}"].shape
            np.testing.assert_allclose(s[i].cpu().numpy(), G[f"rpn_scores{i}"], rtol=0, atol=1e-6)
            np.testing.assert_allclose(b[i].cpu().numpy(), G[f"rpn_boxes{i}"], rtol=0, atol=1e-4)


@pytest.mark.parametrize("n,nseg,live_frac", [(3000, 12, 0.4), (3000, 12, 0.0), (4096, 1024, 0.9), (20000, 364, 0.03), (20000, 364, 1.0),
                                              (150000, 364, 0.5), (70000, 5000, 0.2)])
def test_masked_segmented_nms_equals_compaction(n, nseg, live_frac):
    """tvmi::nms_segmented_masked (round 5: filtered-out candidates leave the problem on the device — score -inf / largest key,
    live count read from memory) == boolean compaction + tvmi::nms_segmented, index for index, on both device-count paths
    (small segments up to 4096 candidates, banded segment-major above), incl. nothing / everything live."""
    g = gen(500 + n + nseg)
    boxes = random_boxes(n, 600, 500, 4, 120, g).to(DEV)
    scores = torch.rand(n, generator=g).to(DEV)
    seg = torch.randint(0, nseg, (n,), generator=g).to(DEV)
    valid = (torch.rand(n, generator=g) < live_frac).to(DEV)
    sel = valid.nonzero()[:, 0]
    want = sel[torch.ops.tvmi.nms_segmented(boxes[sel], scores[sel], seg[sel], 0.5, nseg)] if sel.numel() else sel
    # static per-segment bound: unknown (-1: n <= 4096 then takes the segment-major path, which cannot fail there) and the
    # true one (small path when it is <= 1024)
    bound = int(torch.bincount(seg, minlength=nseg).max())
    for mx in (-1, bound):
        keep, num = torch.ops.tvmi.nms_segmented_masked(boxes, scores, seg, valid.to(torch.uint8), 0.5, nseg, mx)
        assert int(num) == want.numel(), (mx, int(num), want.numel())
        assert torch.equal(keep[: int(num)], want), mx


def test_masked_nms_limits_are_static_and_safe():
    """ADVICE r05 (high / medium / low): the masked op used to return num = -1 for a segment above 1,024 boxes whenever
    n <= 4096 (the same data passed at n = 4097), and the fused post-processing turned that -1 into empty detections.
    Now (a) the small path is taken only under a static bound that fits it, (b) filter_proposals with a single-level RPN
    and pre_nms_top_n = 2000 (2 images: n = 4000 <= 4096, 2000-box segments) equals the oracle, (c) beyond the device-count
    limits (a segment bound above 8,192 / a grid above 1.2 M candidates) the host mirror compacts first, (d) a -1 that
    still reaches the list-returning form raises instead of producing zeros."""
    from vision_amd import detection_post as dp

    g = gen(2000)
    n = 4000
    boxes = random_boxes(n, 900, 700, 8, 200, g).to(DEV)
    scores = torch.rand(n, generator=g).to(DEV)
    seg = (torch.arange(n) // 2000).to(DEV)
    valid = torch.ones(n, dtype=torch.uint8, device=DEV)
    want = torch.ops.tvmi.nms_segmented(boxes, scores, seg, 0.7, 2)
    for mx in (-1, 2000, 4000):
        keep, num = torch.ops.tvmi.nms_segmented_masked(boxes, scores, seg, valid, 0.7, 2, mx)
        assert int(num) == want.numel() and torch.equal(keep[: int(num)], want), mx
    # a wrong promise (bound 1000 for 2000-box segments) is reported, not silently mis-computed
    keep, num = torch.ops.tvmi.nms_segmented_masked(boxes, scores, seg, valid, 0.7, 2, 1000)
    assert int(num) == -1
    with pytest.raises(RuntimeError, match="beyond its static bound"):
        dp._split(torch.zeros(2, 5, 6, device=DEV), torch.tensor([-1, -1], device=DEV), False)
    # (b) single-level RPN, training-size pre_nms_top_n
    B, A = 2, 6000
    shapes = [(600, 800), (576, 768)]
    props = torch.stack([random_boxes(A, 800, 600, 4, 300, g) for _ in range(B)])
    obj = torch.randn(B, A, generator=g)
    kw = dict(pre_nms_top_n=2000, post_nms_top_n=2000, nms_thresh=0.7, score_thresh=0.0)
    got_b, got_s = vision_amd.filter_proposals(props.to(DEV), obj.to(DEV), shapes, [A], **kw)
    ref = O.filter_proposals(props.numpy(), obj.numpy(), shapes, [A], 2000, 2000, 0.7, 0.0, 1e-3)
    for i, (rb, rs) in enumerate(ref):
        assert got_b[i].shape[0] == rb.shape[0] > 200
        np.testing.assert_allclose(got_s[i].cpu().numpy(), rs, rtol=0, atol=1e-6)
        np.testing.assert_allclose(got_b[i].cpu().numpy(), rb, rtol=0, atol=1e-4)
    # (c) the compacting route (forced by lowering the limits) == the device-count route, padded form included
    logits = torch.randn(600, 21, generator=g) * 3
    reg = torch.randn(600, 84, generator=g) * 0.5
    pr = [random_boxes(300, 800, 600, 8, 300, g).to(DEV) for _ in range(2)]
    a = vision_amd.postprocess_detections(logits.to(DEV), reg.to(DEV), pr, shapes, padded=True)
    cap, lim = dp.NMS_CAPACITY, dp.SEGMENT_LIMIT
    try:
        dp.NMS_CAPACITY = 1000
        b = vision_amd.postprocess_detections(logits.to(DEV), reg.to(DEV), pr, shapes, padded=True)
        dp.NMS_CAPACITY, dp.SEGMENT_LIMIT = cap, 100
        c = vision_amd.postprocess_detections(logits.to(DEV), reg.to(DEV), pr, shapes, padded=True)
    finally:
        dp.NMS_CAPACITY, dp.SEGMENT_LIMIT = cap, lim
    assert int(a[1].min()) > 0
    for other in (b, c):
        assert torch.equal(a[0], other[0]) and torch.equal(a[1], other[1])


def test_batched_nms_partition_edge_cases():
    """The radix partition of the segment-major path (round 5) against the reference arithmetic of the oracle: exact score
    ties inside and across segments (stable order), ids that are large, sparse, negative or beyond 2^31 (the partition raises
    its flag and the glue takes the general path — the reference accepts any int64 ids), one box per segment, one segment;
    and the device-count form with a segment above the per-segment limit reporting -1 instead of a wrong list."""
    g = gen(77)
    n = 6000
    boxes = random_boxes(n, 500, 400, 4, 90, g)
    scores = (torch.rand(n, generator=g) * 64).round() / 64          # 65 distinct values: thousands of exact ties
    for name, ids in {
        "dense80": torch.randint(0, 80, (n,), generator=g),
        "sparse_large": torch.randint(0, 50, (n,), generator=g) * 40_000_003 + 7,          # up to ~2e9 < 2^31
        "negative": torch.randint(-20, 20, (n,), generator=g),
        "beyond_int32": torch.randint(0, 30, (n,), generator=g) * (1 << 33),
        "one_per_segment": torch.randperm(n, generator=g),
        "one_segment": torch.zeros(n, dtype=torch.int64),
    }.items():
        want = O.nms(boxes.numpy(), scores.numpy(), 0.5, ids.numpy())
        got = torch.ops.tvmi.nms_segmented(boxes.to(DEV), scores.to(DEV), ids.to(DEV), 0.5, -1).cpu().numpy()
        assert np.array_equal(got, want), name
    # device-count form: 9000 live boxes in ONE segment (limit 8192) -> num = -1, never a silently wrong keep list
    nb = 9500
    b2 = random_boxes(nb, 3000, 3000, 4, 30, g).to(DEV)
    s2 = torch.rand(nb, generator=g).to(DEV)
    valid = torch.ones(nb, dtype=torch.uint8, device=DEV)
    valid[9000:] = 0
    keep, num = torch.ops.tvmi.nms_segmented_masked(b2, s2, torch.zeros(nb, dtype=torch.int64, device=DEV), valid, 0.5, 1)
    assert int(num) == -1
    # ... and an id outside the promised range in the masked form
    ids = torch.randint(0, 5, (nb,), generator=g).to(DEV)
    ids[17] = 9
    keep, num = torch.ops.tvmi.nms_segmented_masked(b2, s2, ids, valid, 0.5, 5)
    assert int(num) == -1
    ids[17] = 3
    ids[9100] = 9                      # the same id on a masked-out candidate does not matter
    keep, num = torch.ops.tvmi.nms_segmented_masked(b2, s2, ids, valid, 0.5, 5)
    sel = valid.bool().nonzero()[:, 0]
    want = sel[torch.ops.tvmi.nms_segmented(b2[sel], s2[sel], ids[sel], 0.5, 5)]
    assert int(num) == want.numel() and torch.equal(keep[: int(num)], want)


def test_detector_postprocessing_is_sync_free_in_padded_form():
    """the three fused post-processing functions (RoIHeads.postprocess_detections roi_heads.py:680-737,
    RegionProposalNetwork.filter_proposals rpn.py:242-297, RetinaNet.postprocess_detections retinanet.py:509-571) run in
    their padded form under torch.cuda.set_sync_debug_mode("error") — no nonzero, no .item(), no blocking copy — and give
    what the list-returning form (one read of the counts at the end) gives."""
    g = gen(93)
    shapes = [(800, 1333), (800, 1200), (750, 1333)]
    B = len(shapes)
    props = [random_boxes(600, w, h, 8, 500, g).to(DEV) for h, w in shapes]
    logits = (torch.randn(B * 600, 91, generator=g) * 3).to(DEV)
    reg = (torch.randn(B * 600, 364, generator=g) * 0.5).to(DEV)
    A_lvls = [3000, 800, 200]
    A = sum(A_lvls)
    anchors = random_boxes(B * A, 1200, 750, 16, 300, g).reshape(B, A, 4).to(DEV)
    obj = torch.randn(B, A, generator=g).to(DEV)
    deltas = (torch.randn(B, A, 4, generator=g) * 0.3).to(DEV)
    K = 20
    cls = [(torch.randn(B, a, K, generator=g) * 2 - 2).to(DEV) for a in A_lvls]
    breg = [(torch.randn(B, a, 4, generator=g) * 0.3).to(DEV) for a in A_lvls]
    off = [0, 3000, 3800, 4000]
    anc_lists = [[anchors[i, off[l]:off[l + 1]] for l in range(3)] for i in range(B)]
    calls = {
        "roi_heads": lambda padded: vision_amd.postprocess_detections(logits, reg, props, shapes, score_thresh=0.05, padded=padded),
        "rpn": lambda padded: vision_amd.filter_proposals(anchors, obj, shapes, A_lvls, pre_nms_top_n=1000, post_nms_top_n=300,
                                                          score_thresh=0.2, pred_bbox_deltas=deltas, padded=padded),
        "retinanet": lambda padded: vision_amd.retinanet_postprocess_detections(cls, breg, anc_lists, shapes, score_thresh=0.05,
                                                                                topk_candidates=500, detections_per_img=100, padded=padded),
    }
    for name, fn in calls.items():
        want = fn(False)
        fn(True)                                   # first use: workspace allocations etc. happen outside the checked region
        torch.cuda.synchronize()
        torch.cuda.set_sync_debug_mode("error")
        try:
            dets, counts = fn(True)
        finally:
            torch.cuda.set_sync_debug_mode("default")
        counts = counts.tolist()
        assert all(c >= 0 for c in counts), (name, counts)
        if name == "retinanet":
            want = ([w["boxes"] for w in want], [w["scores"] for w in want], [w["labels"] for w in want])
        assert sum(counts) > 0, name
        for i in range(B):
            d = dets[i, : counts[i]]
            assert counts[i] == want[0][i].shape[0], (name, i)
            assert torch.equal(d[:, :4], want[0][i]) and torch.equal(d[:, 4], want[1][i]), (name, i)
            if len(want) == 3:
                assert torch.equal(d[:, 5].to(torch.int64), want[2][i]), (name, i)


def test_sync_free_nms_pack_chain_and_graph_replay():
    """batched_nms_padded + pack_kept_detections(num_keep=...) == the synchronising pair, eagerly and when the
    chain is captured in a hipGraph and replayed on fresh input values."""
    from vision_amd import sharding
    g = gen(71)
    n, B = 2700, 3   # ~900 boxes per image: inside the 1024-per-segment limit of the small path
    boxes = random_boxes(n, 800, 600, 4, 200, g).to(DEV)
    scores = torch.rand(n, generator=g).to(DEV)
    img = torch.randint(0, B, (n,), generator=g).to(DEV)
    keep = torch.ops.tvmi.nms_segmented(boxes, scores, img, 0.5, -1)    # the synchronising form of the same per-segment arithmetic
    want_d, want_c = sharding.pack_kept_detections(boxes, scores, img, keep, B, 50)
    kp, num = vision_amd.boxes.batched_nms_padded(boxes, scores, img, 0.5, B)
    assert int(num) == keep.numel() and torch.equal(kp[: keep.numel()], keep)
    got_d, got_c = sharding.pack_kept_detections(boxes, scores, img, kp, B, 50, num_keep=num)
    assert torch.equal(got_d, want_d) and torch.equal(got_c, want_c)
    # graph capture / replay with the inputs overwritten in place
    sb, ss = boxes.clone(), scores.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            k2, n2 = vision_amd.boxes.batched_nms_padded(sb, ss, img, 0.5, B)
            sharding.pack_kept_detections(sb, ss, img, k2, B, 50, num_keep=n2)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        k2, n2 = vision_amd.boxes.batched_nms_padded(sb, ss, img, 0.5, B)
        d2, c2 = sharding.pack_kept_detections(sb, ss, img, k2, B, 50, num_keep=n2)
    nb = random_boxes(n, 800, 600, 4, 200, g).to(DEV)
    ns = torch.rand(n, generator=g).to(DEV)
    sb.copy_(nb)
    ss.copy_(ns)
    graph.replay()
    torch.cuda.synchronize()
    keep = torch.ops.tvmi.nms_segmented(nb, ns, img, 0.5, -1)
    want_d, want_c = sharding.pack_kept_detections(nb, ns, img, keep, B, 50)
    assert int(n2) == keep.numel() and torch.equal(d2, want_d) and torch.equal(c2, want_c)


def test_roi_align_channels_last_native_kernel(tv):
    """channels_last (NHWC) feature maps take the lane=channel kernel; result must equal the NCHW path's (same
    arithmetic) and the oracle, for single-level roi_align (incl. aligned / edge RoIs / C not a multiple of 64) and the
    multi-scale op; output stays NCHW-contiguous like the reference's."""
    g = gen(83)
    for C, aligned in ((96, False), (256, True), (70, False)):
        N, H, W = 2, 46, 61
        x = torch.rand(N, C, H, W, generator=g)
        rois = rois_for(N, 150, W * 8, H * 8, 8, 300, g)
        rois[0, 1:] = torch.tensor([0.0, 0.0, W * 8.0, H * 8.0])
        rois[1, 1:] = torch.tensor([W * 8 - 4.0, H * 8 - 4.0, W * 8 + 30.0, H * 8 + 20.0])
        rois[2, 1:] = torch.tensor([-50.0, -40.0, -20.0, -10.0])   # entirely outside: zeros
        xl = x.to(DEV).contiguous(memory_format=torch.channels_last)
        assert not xl.is_contiguous()
        y = tv.roi_align(xl, rois.to(DEV), 1 / 8, 7, 7, 2, aligned)
        assert y.is_contiguous() and y.shape == (150, C, 7, 7)
        ref = O.roi_align(x.numpy(), rois.numpy(), 1 / 8, 7, 7, 2, aligned)
        np.testing.assert_allclose(y.cpu().numpy(), ref, rtol=0, atol=TOL)
        y2 = tv.roi_align(x.to(DEV), rois.to(DEV), 1 / 8, 7, 7, 2, aligned)
        np.testing.assert_allclose(y.cpu().numpy(), y2.cpu().numpy(), rtol=0, atol=1e-6)
    feats = {str(i): torch.rand(2, 64, 800 // s, 1344 // s, generator=g) for i, s in enumerate((4, 8, 16, 32))}
    boxes = [random_boxes(200, 1344, 800, 8, 700, g) for _ in range(2)]
    pool = vision_amd.MultiScaleRoIAlign(["0", "1", "2", "3"], 7, 2)
    with torch.no_grad():
        a = pool({k: v.to(DEV) for k, v in feats.items()}, [b.to(DEV) for b in boxes], [(800, 1344)] * 2)
        b = pool({k: v.to(DEV).contiguous(memory_format=torch.channels_last) for k, v in feats.items()},
                 [b.to(DEV) for b in boxes], [(800, 1344)] * 2)
    assert b.is_contiguous()
    np.testing.assert_allclose(b.cpu().numpy(), a.cpu().numpy(), rtol=0, atol=1e-6)
    # 16-bit channels_last maps: two channels per lane; same values as the 16-bit NCHW path (fp32 accumulation in both)
    for dt in (torch.float16, torch.bfloat16):
        for C in (64, 200):
            x = torch.rand(2, C, 46, 61, generator=g).to(dt)
            rois = rois_for(2, 120, 61 * 8, 46 * 8, 8, 300, g).to(dt)
            y_cl = tv.roi_align(x.to(DEV).contiguous(memory_format=torch.channels_last), rois.to(DEV), 1 / 8, 7, 7, 2, False)
            y_nc = tv.roi_align(x.to(DEV), rois.to(DEV), 1 / 8, 7, 7, 2, False)
            assert y_cl.dtype == dt and y_cl.is_contiguous()
            assert torch.equal(y_cl, y_nc)


def test_boxes_to_rois_one_launch():
    g = gen(5)
    for dt in (torch.float32, torch.float16, torch.float64):
        lists = [random_boxes(n, 300, 200, 2, 100, g).to(dt) for n in (5, 0, 17, 1)]
        want = torch.cat([torch.cat([torch.full_like(b[:, :1], i) for i, b in enumerate(lists)]), torch.cat(lists)], 1)
        got = vision_amd.roi_ops.convert_boxes_to_roi_format([b.to(DEV) for b in lists])
        assert got.dtype == dt and torch.equal(got.cpu(), want)


def test_multiscale_roi_align_backward_single_launch():
    """Autograd through the fused multi-scale op (one backward launch for all levels) == the per-level loop of
    torchvision/ops/poolers.py:199-222 run through `_roi_align_backward`, and == the oracle level by level."""
    g = gen(97)
    N, C = 2, 40
    shapes = [(800 // s, 1344 // s) for s in (4, 8, 16, 32)]
    feats = [torch.rand(N, C, h, w, generator=g) for h, w in shapes]
    boxes = [random_boxes(150, 1344, 800, 8, 700, g) for _ in range(N)]
    for P in (7, 14, 5):
        pool = vision_amd.MultiScaleRoIAlign(["0", "1", "2", "3"], P, 2)
        fd = [f.to(DEV).requires_grad_(True) for f in feats]
        out = pool({str(i): f for i, f in enumerate(fd)}, [b.to(DEV) for b in boxes], [(800, 1344)] * N)
        gout = torch.randn(out.shape, generator=g)
        out.backward(gout.to(DEV))
        rois = vision_amd.roi_ops.convert_boxes_to_roi_format(boxes)
        levels = pool.map_levels(boxes)
        for l, (h, w) in enumerate(shapes):
            idx = torch.nonzero(levels == l)[:, 0]
            ref = O.roi_align_backward(gout[idx].numpy(), rois[idx].numpy(), pool.scales[l], P, P, N, C, h, w, 2, False)
            np.testing.assert_allclose(fd[l].grad.cpu().numpy(), ref, rtol=1e-4, atol=TOL * max(1.0, float(np.abs(ref).max())))
    # 16-bit maps: fp32 accumulation inside, dtype preserved outside
    fd = [f.to(DEV).half().requires_grad_(True) for f in feats]
    pool = vision_amd.MultiScaleRoIAlign(["0", "1", "2", "3"], 7, 2)
    out = pool({str(i): f for i, f in enumerate(fd)}, [b.to(DEV) for b in boxes], [(800, 1344)] * N)
    out.float().sum().backward()
    assert all(f.grad is not None and f.grad.dtype == torch.float16 for f in fd)


def test_box_iou_pairwise_bit_identical_to_reference_tensor_math():
    """One-launch box_iou / generalized_box_iou vs the reference formulas (ops/boxes.py:314-391, 409-436) evaluated
    with torch CPU tensor math: identical bits (same operations, same order, no contraction), incl. the reference's
    known-answer matrix (test/test_ops.py:1652-1658), degenerate boxes (0/0 = NaN) and float16 upcast."""
    from vision_amd import boxes as VB
    g = gen(17)
    b1 = random_boxes(37, 500, 400, 1, 200, g)
    b2 = random_boxes(1000, 500, 400, 1, 200, g)
    b2[5] = b1[3]
    b2[6, 2:] = b2[6, :2]                      # zero-area box
    for a, b in ((b1, b2), (b1.double(), b2.double()), (b1.half(), b2.half())):
        inter, union = VB._box_inter_union(a, b)          # CPU tensor math == the reference's
        want = inter / union
        got = vision_amd.box_iou(a.to(DEV), b.to(DEV)).cpu()
        assert got.dtype == want.dtype and torch.equal(torch.nan_to_num(got, nan=-7.0), torch.nan_to_num(want, nan=-7.0))
        lti = torch.min(a[:, None, :2], b[None, :, :2]); rbi = torch.max(a[:, None, 2:], b[None, :, 2:])
        whi = VB._upcast(rbi - lti).clamp(min=0); areai = whi[..., 0] * whi[..., 1]
        wantg = want - (areai - union) / areai
        gotg = vision_amd.generalized_box_iou(a.to(DEV), b.to(DEV)).cpu()
        assert torch.equal(torch.nan_to_num(gotg, nan=-7.0), torch.nan_to_num(wantg, nan=-7.0))
    kat1 = torch.tensor([[0, 0, 100, 100], [0, 0, 50, 50], [200, 200, 300, 300], [0, 0, 25, 25]], dtype=torch.float32)
    kat2 = torch.tensor([[0, 0, 100, 100], [0, 0, 50, 50], [200, 200, 300, 300]], dtype=torch.float32)
    expected = torch.tensor([[1.0, 0.25, 0.0], [0.25, 1.0, 0.0], [0.0, 0.0, 1.0], [0.0625, 0.25, 0.0]])
    assert torch.allclose(vision_amd.box_iou(kat1.to(DEV), kat2.to(DEV)).cpu(), expected, atol=1e-4)


def test_distance_and_complete_box_iou_fused_kernel():
    """distance_box_iou / complete_box_iou (ops/boxes.py:439-515) in the fused pairwise kernel vs the same formulas in
    torch CPU tensor math: DIoU is the same operations in the same order (1e-6 covers the device division), CIoU adds
    atan (device libm vs the CPU's: 1e-5); plus the reference's known-answer matrix (test/test_ops.py:1781-1788)."""
    g = gen(23)
    b1 = random_boxes(41, 500, 400, 2, 200, g)
    b2 = random_boxes(700, 500, 400, 2, 200, g)
    b2[3] = b1[7]
    for a, b in ((b1, b2), (b1.double(), b2.double()), (b1.half(), b2.half())):
        for fn, tol in ((vision_amd.distance_box_iou, 1e-6), (vision_amd.complete_box_iou, 1e-5)):
            want = fn(a, b)                                 # CPU tensors: the tensor-math form
            got = fn(a.to(DEV), b.to(DEV)).cpu()
            assert got.dtype == want.dtype and got.shape == (41, 700)
            torch.testing.assert_close(got, want, rtol=0, atol=tol)
    f = torch.tensor([[285.3538, 185.5758, 1193.5110, 851.4551], [285.1472, 188.7374, 1192.4984, 851.0669],
                      [279.2440, 197.9812, 1189.4746, 849.2019]])
    expected = torch.tensor([[1.0, 0.9933, 0.9673], [0.9933, 1.0, 0.9737], [0.9673, 0.9737, 1.0]])
    assert torch.allclose(vision_amd.distance_box_iou(f.to(DEV), f.to(DEV)).cpu(), expected, atol=1e-3)
    assert torch.allclose(vision_amd.complete_box_iou(f.to(DEV), f.to(DEV)).cpu(), expected, atol=1e-3)


def test_transform_images_one_launch_golden():
    """Fused normalize + resize + batching vs the reference GeneralizedRCNNTransform's own output (eval mode): same
    image_sizes and padded shape, values within 2e-5 (fp32 bilinear on normalised taps), padding exactly zero."""
    G = golden("detection")
    imgs = [torch.from_numpy(G[f"xform_img{i}"]).to(DEV) for i in range(4)]
    for tag, kw in (("a", dict(min_size=96, max_size=160)), ("b", dict(min_size=64, max_size=100)),
                    ("c", dict(min_size=50, max_size=80, fixed_size=(72, 56)))):
        out, sizes = vision_amd.transform_images(imgs, **kw)
        want = G[f"xform_{tag}_out"]
        assert [tuple(s) for s in G[f"xform_{tag}_sizes"]] == sizes and tuple(out.shape) == want.shape
        np.testing.assert_allclose(out.cpu().numpy(), want, rtol=0, atol=2e-5)
        assert np.array_equal(out.cpu().numpy() == 0, want == 0)
    # model-sized inputs against the oracle restatement (pinned to torch CPU by tests/test_oracle.py)
    g = gen(43)
    big = [torch.rand(3, h, w, generator=g) for h, w in ((1080, 1920), (480, 640))]
    out, sizes = vision_amd.transform_images([b.to(DEV) for b in big])
    ref, rsizes = O.transform_images([b.numpy() for b in big], 800, 1333, [0.485, 0.456, 0.406], [0.229, 0.224, 0.225])
    assert sizes == rsizes and tuple(out.shape) == ref.shape
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=0, atol=2e-5)


@pytest.mark.parametrize("opname", ["roi_pool", "ps_roi_pool", "roi_align", "ps_roi_align"])
@pytest.mark.parametrize("dtype", [torch.float16, torch.float32, torch.float64])
@pytest.mark.parametrize("requires_grad", [True, False])
def test_roi_opcheck(opname, dtype, requires_grad):
    """The reference's `test_roi_opcheck` (test/test_ops.py:761-795) against OUR schema / fake / autograd
    registrations and kernels: torch.library.opcheck = schema, autograd registration, fake tensor, AOT dispatch."""
    op = getattr(torch.ops.torchvision, opname)
    rois = torch.tensor([[0, 0, 0, 9, 9], [0, 0, 5, 4, 9], [0, 5, 5, 9, 9], [1, 0, 0, 9, 9]], dtype=dtype, device=DEV,
                        requires_grad=requires_grad)
    pool_size = 5
    x = torch.rand(2, 2 * pool_size ** 2, 10, 10, dtype=dtype, device=DEV)
    kwargs = dict(rois=rois, spatial_scale=1, pooled_height=pool_size, pooled_width=pool_size)
    if opname in ("roi_align", "ps_roi_align"):
        kwargs["sampling_ratio"] = -1
    if opname == "roi_align":
        kwargs["aligned"] = True
    torch.library.opcheck(op, args=(x,), kwargs=kwargs)


def test_nms_and_fused_ops_opcheck():
    g = gen(3)
    b = random_boxes(50, 100, 100, 2, 40, g).to(DEV)
    s = torch.rand(50, generator=g).to(DEV)
    torch.library.opcheck(torch.ops.torchvision.nms, args=(b, s, 0.5))
    torch.library.opcheck(torch.ops.tvmi.nms_segmented, args=(b, s, torch.randint(0, 3, (50,), generator=g).to(DEV), 0.5))
    seg3 = torch.randint(0, 3, (50,), generator=g).to(DEV)
    torch.library.opcheck(torch.ops.tvmi.nms_segmented_masked, args=(b, s, seg3, (torch.rand(50, generator=g) < 0.5).to(DEV), 0.5, 3))
    kp, nk = torch.ops.tvmi.nms_segmented_padded(b, s, seg3, 0.5, 3)
    torch.library.opcheck(torch.ops.tvmi.pack_detections_payload, args=(b, s, None, seg3, kp, nk, 3, 10))
    feats = [torch.rand(1, 8, 64 // k, 64 // k, generator=g).to(DEV).requires_grad_(True) for k in (1, 2)]
    rois = torch.cat([torch.zeros(50, 1), random_boxes(50, 64, 64, 2, 40, g)], 1).to(DEV)
    torch.library.opcheck(torch.ops.tvmi.multiscale_roi_align,
                          args=(feats, rois, [1.0, 0.5], 7, 7, 2, False, 0, 1, 224.0, 4.0, 1e-6))
    # the one-launch step (inference entry: no autograd formula, schema + fake kernel + dispatch are what is checked)
    f2 = [f.detach() for f in feats]
    boxes2 = [random_boxes(30, 64, 64, 2, 40, g).to(DEV)]
    torch.library.opcheck(torch.ops.tvmi.roi_align_boxes_nms_step,
                          args=(f2, boxes2, [1.0, 0.5], 7, 7, 2, False, 0, 1, 224.0, 4.0, 1e-6, b, s, seg3, 0.5, 3, seg3, None, 3, 10),
                          test_utils=("test_schema", "test_faketensor"))


def test_small_score_sort_equals_stable_descending_sort():
    """tvmi::sort_scores_desc (what nms uses for <= 4096 float32 scores) == aten::sort(stable=True, descending=True)
    indices: ties by ascending index, NaN first, +-inf, -0 == +0, every size class of the bitonic network."""
    g = gen(11)
    for n in (1, 2, 3, 63, 64, 65, 1000, 2048, 2049, 4096):
        s = torch.randn(n, generator=g)
        s = (s * 8).round() / 8                      # many exact ties
        if n > 10:
            s[3] = float("nan"); s[7] = float("inf"); s[8] = float("-inf"); s[5] = -0.0; s[6] = 0.0; s[9] = float("nan")
        want = torch.sort(s, stable=True, descending=True)[1]
        got = torch.ops.tvmi.sort_scores_desc(s.to(DEV)).cpu()
        assert torch.equal(got, want), n


def test_nms_large_path_from_two_host_threads(tv):
    """Two host threads run the large path on the same GPU at once (own side streams each).  Only one call per device may
    use the device-side hand-offs at a time — polling kernels of two threads on shared hardware queues could otherwise wait
    for each other — the other one takes the stream-event form; both must finish and give the single-thread index lists."""
    import threading
    g = gen(41)
    cases = [(random_boxes(n, c, c, 1, 101, g).to(DEV), torch.rand(n, generator=g).to(DEV)) for n, c in ((30_000, 400), (45_000, 900))]
    want = [tv.nms(b, s, 0.5).cpu() for b, s in cases]
    errors = []

    def worker(i):
        try:
            b, s = cases[i]
            for _ in range(6):
                if not torch.equal(tv.nms(b, s, 0.5).cpu(), want[i]):
                    errors.append(("mismatch", i))
        except Exception as exc:   # pragma: no cover - the assertion below reports it
            errors.append((repr(exc), i))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t_ in threads:
        t_.start()
    for t_ in threads:
        t_.join(timeout=120)
    assert not any(t_.is_alive() for t_ in threads), "a thread is stuck in nms"
    assert not errors, errors


def test_deform_conv2d_channels_last_gather_is_bit_identical():
    """The 16-bit MFMA kernel samples a [B, H*W, C] copy of the input when the channel counts allow 16-byte octets
    (dcn.channels_last_gather, 32-deep K slabs): same corner values, same rounding, exact products, fp32 sums grouped in
    16s from the start of every offset-group segment => the same bits as the planar-gather kernel when the channels per
    offset group are a multiple of 16, and the same values up to the order of fp32 additions otherwise; every workgroup
    tile, offset groups that cut the slabs, weight groups, stride / dilation, mask on and off; shapes whose channel counts
    are not multiples of 8 silently keep the planar kernel."""
    g = gen(77)
    opt = torch.ops.tvmi.set_option
    try:
        for dt in (torch.bfloat16, torch.float16):
            for (B, C, H, W, OC, groups, ogroups, stride, dil) in ((2, 64, 23, 31, 256, 1, 1, 1, 1), (1, 48, 17, 19, 160, 1, 2, 2, 1),
                                                                  (2, 144, 16, 21, 96, 2, 6, 1, 2), (1, 40, 9, 11, 32, 1, 5, 1, 1),
                                                                  (1, 20, 9, 11, 32, 1, 5, 1, 1), (1, 160, 12, 13, 192, 1, 2, 1, 1)):
                x = torch.randn(B, C, H, W, generator=g).to(DEV, dt)
                w = (torch.randn(OC, C // groups, 3, 3, generator=g) * 0.1).to(DEV, dt)
                oh, ow = (H - 1) // stride + 1, (W - 1) // stride + 1
                off = (torch.randn(B, 2 * 9 * ogroups, oh, ow, generator=g) * 3).to(DEV, dt)
                msk = torch.rand(B, 9 * ogroups, oh, ow, generator=g).to(DEV, dt)
                bias = torch.randn(OC, generator=g).to(DEV, dt)
                for m in (None, msk):
                    opt("dcn.channels_last_gather", 0)
                    want = vision_amd.deform_conv2d(x, off, w, bias, stride=stride, padding=dil, dilation=dil, mask=m)
                    opt("dcn.channels_last_gather", 1)
                    # the pipelined kernel (1) and the non-pipelined one that serves copies of 4 GB and more (0), XCD tile dealing on / off
                    for variant, xcd in ((1, 1), (1, 0), (0, 1)):
                        opt("dcn.cl_variant", variant)
                        opt("dcn.xcd_tiles", xcd)
                        got = vision_amd.deform_conv2d(x, off, w, bias, stride=stride, padding=dil, dilation=dil, mask=m)
                        if (C // ogroups) % 16 == 0 or C % 8 != 0:
                            assert torch.equal(got, want), (dt, C, OC, groups, ogroups, m is not None, variant, xcd)
                        else:   # same products, another grouping of the fp32 additions, then one rounding to 16 bits
                            err = (got.float() - want.float()).abs().max().item()
                            assert err <= 2.0 ** (-7 if dt == torch.bfloat16 else -10) * want.float().abs().max().item(), (dt, C, err)
    finally:
        opt("dcn.channels_last_gather", 1)
        opt("dcn.cl_variant", 1)
        opt("dcn.xcd_tiles", 1)


def test_large_score_sort_equals_stable_descending_sort():
    """Above 4096 scores the order comes from score_sort.hip (key pass + radix sort of (key, index) pairs): same
    contract — ties by ascending index, NaN (any sign / payload) first, +-inf, -0 == +0, denormals."""
    g = gen(12)
    for n in (4097, 10_000, 100_000, 300_001):
        s = torch.randn(n, generator=g)
        s = (s * 64).round() / 64                    # many exact ties
        s[3] = float("nan"); s[7] = float("inf"); s[8] = float("-inf"); s[5] = -0.0; s[6] = 0.0; s[9] = float("nan")
        s[n - 1] = -0.0; s[n - 2] = 0.0; s[n // 2] = float("nan"); s[11] = 1e-42; s[12] = -1e-42
        s[13] = torch.tensor(-0x400000, dtype=torch.int32).view(torch.float32)   # a negative NaN with another payload
        want = torch.sort(s, stable=True, descending=True)[1]
        got = torch.ops.tvmi.sort_scores_desc(s.to(DEV)).cpu()
        assert torch.equal(got, want), n


def test_nms_replanning_on_survivors_is_invisible(tv):
    """torchvision::nms re-plans large problems on the boxes the first chunk(s) left alive (tvmi_nms_blocking): the index
    list must be the one of the single-pass pipeline (re-planning off) for every depth / first-phase size, with and
    without segment ids on the global path, for survivor counts that end in the one-workgroup sweep, and for a list
    whose tail is wiped out completely; one size also against the oracle."""
    g = gen(33)
    opt = torch.ops.tvmi.set_option
    cases = []
    for n, canvas in ((30_011, 300), (30_011, 2000), (50_000, 150)):
        cases.append((random_boxes(n, canvas, canvas, 1, 101, g), torch.rand(n, generator=g), None))
    b = random_boxes(26_000, 400, 400, 5, 80, g)
    cases.append((b, torch.rand(26_000, generator=g), torch.randint(0, 3, (26_000,), generator=g)))
    # a handful of big high-scoring boxes that cover everything: no survivor after the first chunk
    big = torch.tensor([[0.0, 0.0, 100.0, 100.0]]).repeat(5000, 1)
    small = torch.tensor([[0.0, 0.0, 100.0, 99.0]]).repeat(25_000, 1)
    cases.append((torch.cat([big, small]), torch.cat([torch.linspace(1.0, 0.9, 5000), torch.rand(25_000, generator=g) * 0.5]), None))
    try:
        for boxes, scores, idxs in cases:
            bd, sd = boxes.to(DEV), scores.to(DEV)
            call = (lambda: tv.nms(bd, sd, 0.5)) if idxs is None else (lambda: torch.ops.tvmi.nms_segmented(bd, sd, idxs.to(DEV), 0.5, 0))
            opt("nms.replan_min_boxes", 0)
            want = call().cpu()
            for min_boxes, divisor, depth, handoff in ((8192, 8, 1, 1), (8192, 4, 3, 0), (4097, 100, 6, 1), (16384, 2, 2, 1), (0, 8, 1, 0)):
                opt("nms.replan_min_boxes", min_boxes); opt("nms.replan_divisor", divisor); opt("nms.replan_max", depth)
                opt("nms.device_handoff", handoff)   # resolver <-> push hand-offs through memory words / through stream events
                assert torch.equal(call().cpu(), want), (boxes.shape[0], min_boxes, divisor, depth, handoff)
        boxes, scores, _ = cases[0]
        assert np.array_equal(tv.nms(boxes.to(DEV), scores.to(DEV), 0.5).cpu().numpy(), O.nms(boxes.numpy(), scores.numpy(), 0.5))
    finally:
        opt("nms.replan_min_boxes", 24576); opt("nms.replan_divisor", 16); opt("nms.replan_max", 3); opt("nms.device_handoff", 1)


# ------------------------------------------------------------------ tile-owner RoIAlign backward (deterministic)
@pytest.mark.parametrize("P,sr,aligned", [(7, 2, False), (14, 2, True), (7, 0, False), (14, 3, False)])
def test_roi_align_backward_owner_path_layouts(tv, P, sr, aligned):
    """The tile-owner backward (fp32, 7x7 / 14x14): RoI lists longer than one scan round, unsorted batch indices, images
    without RoIs, windows above 64 rows / columns (factors evaluated in the kernel), odd map sizes, adaptive sampling —
    against the oracle; and the result must be bit-identical from run to run."""
    g = gen(200 + P + sr)
    N, C, H, W = 5, 37, 45, 83            # W odd, H / W not multiples of the 16-pixel tile, C not a multiple of 32
    k = 2600
    rois = rois_for(N, k, W * 8, H * 8, 4, 260, g)
    rois[:, 0] = torch.randint(0, N, (k,), generator=g).float()
    rois[rois[:, 0] == 3, 0] = 1.0                                   # image 3 has no RoI at all
    rois[:1500, 1:] = torch.tensor([200.0, 120.0, 330.0, 250.0]) + torch.rand(1500, 4, generator=g) * 6   # > 1024 RoIs on one tile
    rois[:1500, 0] = 2.0
    rois[1500, 1:] = torch.tensor([-40.0, -30.0, 700.0, 400.0])     # window ~ 90 x 54 px
    rois[1501, 1:] = torch.tensor([0.0, 0.0, W * 8.0, H * 8.0])     # the whole map
    rois[1502, 1:] = torch.tensor([300.0, 100.0, 100.0, 50.0])      # malformed (x2 < x1)
    rois[1503, 1:] = torch.tensor([5000.0, 5000.0, 6000.0, 6000.0]) # outside
    gr = torch.randn(k, C, P, P, generator=g)
    a = tv._roi_align_backward(gr.to(DEV), rois.to(DEV), 1 / 8, P, P, N, C, H, W, sr, aligned)
    b = tv._roi_align_backward(gr.to(DEV), rois.to(DEV), 1 / 8, P, P, N, C, H, W, sr, aligned)
    assert torch.equal(a, b)
    ref = O.roi_align_backward(gr.numpy(), rois.numpy(), 1 / 8, P, P, N, C, H, W, sr, aligned)
    np.testing.assert_allclose(a.cpu().numpy(), ref, rtol=2e-4, atol=2e-4 * max(1.0, float(np.abs(ref).max())))
    assert a[3].abs().max().item() == 0.0
    # expanded (stride-0) and transposed gradients still arrive (the glue makes the bins contiguous)
    ones = torch.ones(1, device=DEV).expand(k, C, P, P)
    c = tv._roi_align_backward(ones, rois.to(DEV), 1 / 8, P, P, N, C, H, W, sr, aligned)
    refc = O.roi_align_backward(np.ones((k, C, P, P), np.float32), rois.numpy(), 1 / 8, P, P, N, C, H, W, sr, aligned)
    np.testing.assert_allclose(c.cpu().numpy(), refc, rtol=2e-4, atol=2e-4 * max(1.0, float(np.abs(refc).max())))


@pytest.mark.parametrize("nbig", [5, 1300])
def test_roi_align_backward_owner_oversized_windows(tv, nbig):
    """Windows above the 128-row / -column coefficient tables take the second pass (factors evaluated in the kernel, tile
    read-add-written): a handful of them (sorted short list) and more than the list holds (full descriptor scan)."""
    g = gen(260 + nbig)
    N, C, H, W, P = 2, 6, 150, 171, 7
    k = nbig + 300
    rois = rois_for(N, k, W * 4, H * 4, 8, 200, g)
    big = torch.rand(nbig, 4, generator=g)
    rois[:nbig, 1] = big[:, 0] * 60
    rois[:nbig, 2] = big[:, 1] * 40
    rois[:nbig, 3] = rois[:nbig, 1] + 540 + big[:, 2] * 80          # 135..155 px wide on the map
    rois[:nbig, 4] = rois[:nbig, 2] + 200 + big[:, 3] * 380         # 50..145 px tall
    rois = rois[torch.randperm(k, generator=g)]
    gr = torch.randn(k, C, P, P, generator=g)
    a = tv._roi_align_backward(gr.to(DEV), rois.to(DEV), 0.25, P, P, N, C, H, W, 2, False)
    b = tv._roi_align_backward(gr.to(DEV), rois.to(DEV), 0.25, P, P, N, C, H, W, 2, False)
    assert torch.equal(a, b)
    ref = O.roi_align_backward(gr.numpy(), rois.numpy(), 0.25, P, P, N, C, H, W, 2, False)
    np.testing.assert_allclose(a.cpu().numpy(), ref, rtol=2e-4, atol=2e-4 * max(1.0, float(np.abs(ref).max())))


def test_roi_align_backward_is_deterministic_under_the_torch_flag(tv):
    """torch.use_deterministic_algorithms(True): the reference reroutes roi_align to a python implementation
    (torchvision/ops/roi_align.py:276-281) because its CUDA backward is atomic; ours is deterministic for the detector
    shapes and must therefore not raise, and must raise for the shapes that still accumulate atomically."""
    g = gen(231)
    x = torch.randn(2, 16, 40, 60, generator=g).to(DEV)
    rois = rois_for(2, 300, 480, 320, 8, 200, g).to(DEV)
    feats = {str(i): torch.randn(2, 16, 320 // s, 480 // s, generator=g).to(DEV).requires_grad_(True) for i, s in enumerate((4, 8, 16))}
    boxes = [random_boxes(200, 480, 320, 8, 300, g).to(DEV) for _ in range(2)]
    pool = vision_amd.MultiScaleRoIAlign(["0", "1", "2"], 7, 2)
    torch.use_deterministic_algorithms(True)
    try:
        xa = x.clone().requires_grad_(True)
        vision_amd.roi_align(xa, rois, 7, 1 / 8, 2, False).square().sum().backward()
        xb = x.clone().requires_grad_(True)
        vision_amd.roi_align(xb, rois, 7, 1 / 8, 2, False).square().sum().backward()
        assert torch.equal(xa.grad, xb.grad)
        pool(feats, boxes, [(320, 480)] * 2).square().sum().backward()
        g1 = [f.grad.clone() for f in feats.values()]
        for f in feats.values():
            f.grad = None
        pool(feats, boxes, [(320, 480)] * 2).square().sum().backward()
        assert all(torch.equal(a, f.grad) for a, f in zip(g1, feats.values()))
        xc = x.clone().requires_grad_(True)
        with pytest.raises(RuntimeError, match="roi_align_backward_kernel"):
            vision_amd.roi_align(xc, rois, 5, 1 / 8, 2, False).sum().backward()
    finally:
        torch.use_deterministic_algorithms(False)


def test_multiscale_roi_align_16bit_features_keep_fp32_boxes():
    """fp16 / bf16 feature maps with fp32 proposals: levels and sample coordinates must come from the fp32 boxes (the
    reference casts the RoIs to fp32 under autocast, _autograd_registrations.py:246) — boxes above 1024 px would move by
    up to 8 px if they were rounded to bf16."""
    g = gen(241)
    B, C = 2, 32
    feats = {str(i): torch.randn(B, C, 800 // s, 1344 // s, generator=g) for i, s in enumerate((4, 8, 16, 32))}
    boxes = [random_boxes(200, 1344, 800, 8, 500, g) + torch.tensor([0.37, 0.21, 0.37, 0.21]) for _ in range(B)]
    for b in boxes:
        b[:100, 0::2] += 600.0                  # plenty of coordinates above 1024
        b[:, 2].clamp_(max=1344.0)
    pool = vision_amd.MultiScaleRoIAlign(["0", "1", "2", "3"], 7, 2)
    levels = pool.map_levels(boxes) if pool.map_levels else None
    for dt, tol in ((torch.bfloat16, 5e-3), (torch.float16, 4e-3)):
        with torch.no_grad():
            out = pool({k: v.to(dt).to(DEV) for k, v in feats.items()}, [b.to(DEV) for b in boxes], [(800, 1344)] * B)
        assert out.dtype == dt
        levels = pool.map_levels(boxes)
        rois = torch.cat([torch.cat([torch.full((len(b), 1), float(i)), b], 1) for i, b in enumerate(boxes)])
        for lvl in range(4):
            sel = torch.nonzero(levels == lvl)[:, 0]
            if sel.numel():
                ref = O.roi_align(feats[str(lvl)].to(dt).float().numpy(), rois[sel].numpy(), pool.scales[lvl], 7, 7, 2, False)
                np.testing.assert_allclose(out[sel.to(DEV)].float().cpu().numpy(), ref, rtol=tol, atol=tol)
        # the autocast route gives the same tensor
        with torch.no_grad(), torch.autocast("cuda", dtype=dt):
            out2 = pool({k: v.to(dt).to(DEV) for k, v in feats.items()}, [b.to(DEV) for b in boxes], [(800, 1344)] * B)
        assert torch.equal(out, out2)


def test_pack_kept_payload_is_the_padded_payload_in_place():
    """`tvmi_pack_detections_payload`: the [B, D*6+1] collective payload written by the packing launch
    itself equals what all_gather_detections used to assemble from (dets, counts) with six torch launches; the split is views."""
    from vision_amd import sharding

    g = gen(321)
    B, n, D = 3, 900, 40
    boxes = random_boxes(n, 300, 300, 10, 80, g).to(DEV)
    scores = torch.rand(n, generator=g).to(DEV)
    labels = torch.randint(1, 91, (n,), generator=g).to(DEV)
    img = torch.randint(0, B, (n,), generator=g).to(DEV)
    kp, num = vision_amd.boxes.batched_nms_padded(boxes, scores, img, 0.5, B)
    want_d, want_c = sharding.pack_kept_detections(boxes, scores, img, kp, B, D, labels=labels, num_keep=num)
    payload = sharding.pack_kept_payload(boxes, scores, img, kp, num, B, D, labels=labels)
    assert tuple(payload.shape) == (B, D * 6 + 1) and payload.dtype == torch.float32
    d, c = sharding.split_payload(payload, D)
    assert d.data_ptr() == payload.data_ptr() and torch.equal(d, want_d) and torch.equal(c.round().to(torch.int32), want_c)
    gd, gc = sharding.all_gather_payload(payload, D)          # no process group: views of the same payload
    assert gd.data_ptr() == payload.data_ptr() and torch.equal(gd, want_d)
    out = sharding.unpack_detections(gd, gc)
    assert [o["boxes"].shape[0] for o in out] == want_c.tolist()


@pytest.mark.parametrize("n,ncat", [(900, 5), (6000, 12), (20000, 80)])
def test_batched_nms_mirrors_the_coordinate_trick_with_negative_boxes(n, ncat):
    """VERDICT r03 weak 1a.  In the coordinate-trick regime (<= 100,000 elements on device tensors, ops/boxes.py:83) the
    reference shifts category c by c * (max + 1); with coordinates below -1 the shifted categories overlap and its nms()
    suppresses ACROSS categories (ops/boxes.py:93-109).  The mirror must return the reference's index list there too:
    expected = the reference formulation itself on the CPU (shifted boxes -> the reference CPU nms kernel / its C restatement)."""
    g = gen(4000 + n)
    boxes = random_boxes(n, 160, 160, 4, 90, g) - 70.0          # coordinates in [-70, 90]: min < -1
    boxes[: n // 3] += 70.0                                     # a third stays non-negative
    scores = torch.rand(n, generator=g)
    idxs = torch.randint(0, ncat, (n,), generator=g)
    # planted cross-category pairs: box 0 pins the maximum at 95 (shift per category = 96); A in category c with all
    # coordinates >= 30, B = A - 96 (+ jitter) in category c + 1 with a lower score: after the shifts B lies on A
    boxes[0] = torch.tensor([60.0, 60.0, 95.0, 95.0])
    npairs = 40
    for k in range(npairs):
        a, bidx = 1 + 2 * k, 2 + 2 * k
        xy = 30.0 + torch.rand(2, generator=g) * 20.0
        wh = 20.0 + torch.rand(2, generator=g) * 20.0
        boxes[a] = torch.cat([xy, xy + wh])
        boxes[bidx] = boxes[a] - 96.0 + (torch.rand(4, generator=g) - 0.5)
        c = int(torch.randint(0, ncat - 1, (1,), generator=g))
        idxs[a], idxs[bidx] = c, c + 1
        scores[a], scores[bidx] = 0.99 - 0.001 * k, 0.5 - 0.001 * k
    boxes.clamp_(max=95.0)
    assert float(boxes.max()) == 95.0 and float(boxes.min()) < -1.0
    shifted = boxes + (idxs.to(boxes) * (boxes.max() + 1))[:, None]
    want = O.nms(shifted.numpy(), scores.numpy(), 0.5)
    per_cat = O.nms(boxes.numpy(), scores.numpy(), 0.5, idxs.numpy())
    assert not np.array_equal(want, per_cat), "the case must really suppress across categories"
    got = vision_amd.batched_nms(boxes.to(DEV), scores.to(DEV), idxs.to(DEV), 0.5).cpu().numpy()
    assert np.array_equal(got, want)
    # the same boxes moved into the non-negative range: no overlap of shifted categories, segment-major path, same rule
    pos = boxes + 70.0
    shifted = pos + (idxs.to(pos) * (pos.max() + 1))[:, None]
    got = vision_amd.batched_nms(pos.to(DEV), scores.to(DEV), idxs.to(DEV), 0.5).cpu().numpy()
    assert np.array_equal(got, O.nms(shifted.numpy(), scores.numpy(), 0.5))


def test_batched_nms_fp16_and_many_categories_follow_the_reference_arithmetic():
    """the shift idxs * (max + 1) is taken in the boxes' dtype like the reference (`idxs.to(boxes)`): fp16 boxes
    round / overflow exactly as there.  Expected: the reference formulation on the CPU in fp16 (shifted fp16 boxes widened to
    fp32 for the reference kernel, which evaluates fp16 boxes in fp32 as well — cuda/nms_kernel.cu:32-53)."""
    g = gen(77)
    n = 3000
    for ncat, extent in ((40, 300.0), (2000, 600.0)):          # 2000 * 601 overflows fp16 (max 65504): inf coordinates
        boxes = random_boxes(n, extent, extent, 8, 120, g).half()
        scores = torch.rand(n, generator=g)
        idxs = torch.randint(0, ncat, (n,), generator=g)
        shifted = boxes + (idxs.to(boxes) * (boxes.max() + torch.tensor(1).to(boxes)))[:, None]
        want = O.nms(shifted.float().numpy(), scores.numpy(), 0.5)
        got = vision_amd.batched_nms(boxes.to(DEV), scores.to(DEV), idxs.to(DEV), 0.5).cpu().numpy()
        assert np.array_equal(got, want), (ncat, extent)


def test_nms_large_path_handoff_gives_up_and_recovers(tv):
    """VERDICT r03 item 4 / ADVICE r03: the device-side hand-offs of the large path used to end in __builtin_trap() when a poll
    outlived its bound.  With the test hook `nms.handoff_lose_flag` the resolver of the first chunk never announces itself —
    exactly what the push kernel sees when a tool serialises kernels in the wrong order — so a poll really gives up (about a
    second), the call re-runs itself with stream events and returns the reference's index list; the process then stops using
    the hand-offs (the option reads back 0) until it is switched on again."""
    g = gen(515)
    n = 30_000
    b = random_boxes(n, 500, 500, 1, 101, g)
    s = torch.rand(n, generator=g)
    want = torch.ops.torchvision.nms(b, s, 0.5).numpy() if O.load_reference() else O.nms(b.numpy(), s.numpy(), 0.5)
    opt, get = torch.ops.tvmi.set_option, torch.ops.tvmi.get_option
    try:
        opt("nms.device_handoff", 1)
        if get("nms.device_handoff") != 1:
            pytest.skip("the environment announces serialised kernels: the hand-offs are never taken here")
        opt("nms.handoff_lose_flag", 1)
        got = tv.nms(b.to(DEV), s.to(DEV), 0.5).cpu().numpy()
        assert np.array_equal(got, want)
        assert get("nms.device_handoff") == 0, "a poll that gave up must switch the hand-offs off for the process"
        opt("nms.handoff_lose_flag", 0)
        got = tv.nms(b.to(DEV), s.to(DEV), 0.5).cpu().numpy()      # event form
        assert np.array_equal(got, want)
    finally:
        opt("nms.handoff_lose_flag", 0)
        opt("nms.device_handoff", 1)                                # re-arm
    assert get("nms.device_handoff") == 1
    assert np.array_equal(tv.nms(b.to(DEV), s.to(DEV), 0.5).cpu().numpy(), want)


@pytest.mark.parametrize("var", ["HIP_LAUNCH_BLOCKING", "AMD_SERIALIZE_KERNEL"])
def test_nms_large_path_under_serialising_environment(var):
    """`HIP_LAUNCH_BLOCKING=1` / `AMD_SERIALIZE_KERNEL=3` in a fresh process with the hand-offs switched on: the large path
    (n > 4096) must return the oracle's list (the environment is honoured: stream events), never a GPU fault."""
    import subprocess
    import sys

    from helpers import ROOT

    code = (
        "import sys, numpy as np, torch\n"
        f"sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {ROOT!r} + '/tests')\n"
        "import vision_amd\n"
        "from oracle import oracle as O\n"
        "from helpers import gen, random_boxes\n"
        "torch.ops.tvmi.set_option('nms.device_handoff', 1)\n"
        "g = gen(9); n = 20000\n"
        "b = random_boxes(n, 400, 400, 1, 101, g); s = torch.rand(n, generator=g)\n"
        "got = torch.ops.torchvision.nms(b.cuda(), s.cuda(), 0.5).cpu().numpy()\n"
        "assert np.array_equal(got, O.nms(b.numpy(), s.numpy(), 0.5))\n"
        "print('OK', torch.ops.tvmi.get_option('nms.device_handoff'))\n"
    )
    env = dict(os.environ, **{var: "3" if var == "AMD_SERIALIZE_KERNEL" else "1"})
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0 and "OK" in p.stdout, (p.stdout[-500:], p.stderr[-1500:])


@pytest.mark.parametrize("dtype", [torch.uint8, torch.int8, torch.int16, torch.int32, torch.int64])
def test_qnms_and_qroi_align_device_kernels(tv, dtype):
    """SURVEY.md section 8f-4 (the last item): torchvision::qnms / torchvision::qroi_align on device tensors — CPU-only ops in the
    reference (quantized/cpu/qnms_kernel.cpp, qroi_align_kernel.cpp), integer tensors with explicit (scale, zero point).  Bit-exact
    against the reference CPU kernels (oracle/_ref) or, without them, the numpy restatement on a smaller case."""
    g = gen(90)
    info = torch.iinfo(dtype)
    have_ref = O.load_reference()
    n = 3000 if have_ref else 400
    b = torch.randint(0, min(info.max, 120) - 40, (n, 2), generator=g)
    boxes = torch.cat([b, b + torch.randint(1, 40, (n, 2), generator=g)], 1).to(dtype)
    scores = torch.randint(max(info.min, -100), min(info.max, 127), (n,), generator=g).to(dtype)
    for thr in (0.3, 0.5):
        want = tv.qnms(boxes, scores, thr).numpy() if have_ref else O.qnms(boxes.numpy(), scores.numpy(), thr)
        got = tv.qnms(boxes.to(DEV), scores.to(DEV), thr)
        assert got.dtype == torch.int64 and np.array_equal(got.cpu().numpy(), want), (dtype, thr)
    assert tv.qnms(boxes[:0].to(DEV), scores[:0].to(DEV), 0.5).numel() == 0
    with pytest.raises(RuntimeError, match="same type"):
        tv.qnms(boxes.to(DEV), scores.to(torch.int64 if dtype != torch.int64 else torch.int32).to(DEV), 0.5)
    K, C, H, W = (200, 16, 40, 56) if have_ref else (6, 2, 11, 13)
    lo, hi = max(info.min, -100), min(info.max, 200)
    x = torch.randint(lo, hi + 1, (1, C, H, W), generator=g).to(dtype)
    x1 = torch.randint(0, 2 * W, (K,), generator=g)
    y1 = torch.randint(0, 2 * H, (K,), generator=g)
    rois = torch.stack([torch.zeros(K, dtype=torch.int64), x1, y1, x1 + torch.randint(0, 2 * W, (K,), generator=g),
                        y1 + torch.randint(0, 2 * H, (K,), generator=g)], 1).clamp(max=min(info.max, 120)).to(dtype)
    zp = 3 if info.min < 0 else 120
    for (P, sr, aligned, scale) in ((7, 2, False, 1.0), (3, 0, True, 0.5), (5, 3, False, 0.25)):
        args = (0.07, zp, 0.5, 0, scale, P, P + 1, sr, aligned)
        want = tv.qroi_align(x, rois, *args).numpy() if have_ref else O.qroi_align(x.numpy(), rois.numpy(), *args)
        got = tv.qroi_align(x.to(DEV), rois.to(DEV), *args)
        assert got.dtype == dtype and tuple(got.shape) == (K, C, P, P + 1)
        np.testing.assert_array_equal(got.cpu().numpy(), want, err_msg=f"{dtype} P={P} sr={sr} aligned={aligned}")
    with pytest.raises(RuntimeError, match="one image per batch"):
        tv.qroi_align(torch.cat([x, x]).to(DEV), rois.to(DEV), 0.07, zp, 0.5, 0, 1.0, 3, 3, 2, False)


# ------------------------------------------------------------------ round 6: the detector step's NMS + payload as ONE launch
def _step_case(n, S, B, g, kind):
    if kind == "dense":       # clustered boxes: most are suppressed, long serial chains inside the diagonal tiles
        centers = torch.rand(40, 2, generator=g) * 600
        xy = centers[torch.randint(0, 40, (n,), generator=g)] + torch.randn(n, 2, generator=g) * 10
        wh = 100 + torch.randn(n, 2, generator=g).abs() * 25
        boxes = torch.cat([xy, xy + wh], 1)
    else:
        boxes = random_boxes(n, 1344, 800, 8, 300, g)
    scores = torch.rand(n, generator=g)
    if kind == "ties":
        scores = (scores * 16).floor() / 16          # 17 distinct values: the order is decided by the index
    seg = torch.randint(0, S, (n,), generator=g)
    if kind == "by_image" or int(torch.bincount(seg, minlength=S).max()) > 1024:
        seg = (torch.arange(n) * S // n).to(torch.int64)   # bench.py's layout: segment = image, contiguous runs (<= 1024 each)
        if kind != "by_image":
            seg = seg[torch.randperm(n, generator=g)]       # ... shuffled: equal-sized segments, members scattered
    img = seg if B == S else seg % B
    labels = torch.randint(1, 91, (n,), generator=g)
    return boxes, scores, seg, img, labels


@pytest.mark.parametrize("n,S,B,kind", [(4000, 4, 4, "by_image"), (4000, 4, 4, "dense"), (4096, 64, 16, "random"), (4096, 8, 2, "ties"),
                                        (1024, 1, 1, "random"), (1000, 1, 1, "dense"), (70, 3, 3, "random"), (1, 1, 1, "random"),
                                        (65, 2, 1, "ties"), (3000, 40, 8, "dense"), (2500, 5, 5, "ties")])
def test_nms_step_one_launch_equals_reference_and_chain(n, S, B, kind):
    """tvmi::nms_step (ONE launch: score order, per-segment tiles, sweeps, global-order keep list, padded top-k payload) against
    (a) the oracle restatement of the reference's batched_nms — index list identical, bit for bit — and (b) this library's own
    launch chain for the same sizes (nms_segmented_padded with nms.step_fused = 0, then pack_detections_payload): keep / num /
    payload equal.  Also twice in a row: the hand-over words are re-armed by every call."""
    g = gen(9000 + n + 7 * S + len(kind))
    boxes, scores, seg, img, labels = _step_case(n, S, B, g, kind)
    want = O.nms(boxes.numpy(), scores.numpy(), 0.5, seg.numpy())
    db, ds, dg, di, dl = boxes.to(DEV), scores.to(DEV), seg.to(DEV), img.to(DEV), labels.to(DEV)
    D = 100
    for _ in range(2):
        keep, num, payload = torch.ops.tvmi.nms_step(db, ds, dg, 0.5, S, di, dl, B, D)
        assert int(num) == len(want), (int(num), len(want))
        assert np.array_equal(keep[: int(num)].cpu().numpy(), want)
    torch.ops.tvmi.set_option("nms.step_fused", 0)
    try:
        k2, n2 = torch.ops.tvmi.nms_segmented_padded(db, ds, dg, 0.5, S)
    finally:
        torch.ops.tvmi.set_option("nms.step_fused", 1)
    assert int(n2) == int(num) and torch.equal(k2[: int(n2)], keep[: int(num)])
    p2 = torch.ops.tvmi.pack_detections_payload(db, ds, dl, di, k2, n2, B, D)
    assert torch.equal(payload, p2)
    # the padded form now takes the one-launch kernel by itself, and so does torchvision::nms up to 1024 boxes
    k3, n3 = torch.ops.tvmi.nms_segmented_padded(db, ds, dg, 0.5, S)
    assert int(n3) == int(num) and torch.equal(k3[: int(n3)], keep[: int(num)])
    if S == 1:
        assert np.array_equal(torch.ops.torchvision.nms(db, ds, 0.5).cpu().numpy(), want)


def test_nms_step_limits_and_special_scores():
    """A segment above 1,024 boxes or an id outside [0, S) gives num = -1 and counts of -1 in the payload (never a wrong list); the
    list-returning ops fall back to the general path on it; NaN / inf / -0 scores order like aten::sort (NaN first)."""
    g = gen(424242)
    n = 3000
    boxes = random_boxes(n, 1000, 800, 8, 200, g).to(DEV)
    scores = torch.rand(n, generator=g).to(DEV)
    seg = torch.zeros(n, dtype=torch.int64, device=DEV)
    seg[:100] = 1
    keep, num, payload = torch.ops.tvmi.nms_step(boxes, scores, seg, 0.5, 2, seg, None, 2, 10)
    assert int(num) == -1 and payload[:, -1].tolist() == [-1.0, -1.0] and not payload[:, :-1].any()
    want = O.nms(boxes.cpu().numpy(), scores.cpu().numpy(), 0.5, seg.cpu().numpy())
    got = torch.ops.tvmi.nms_segmented(boxes, scores, seg, 0.5, 2).cpu().numpy()
    assert np.array_equal(got, want)
    seg2 = torch.randint(0, 4, (n,), generator=g).to(DEV)
    seg2[5] = 7
    _, num, _ = torch.ops.tvmi.nms_step(boxes, scores, seg2, 0.5, 4, seg2.clamp(max=3), None, 4, 10)
    assert int(num) == -1
    s3 = scores.clone().cpu()
    s3[3], s3[10], s3[11], s3[12], s3[13] = float("nan"), float("inf"), float("-inf"), -0.0, 0.0
    seg3 = torch.randint(0, 4, (n,), generator=g)
    want = O.nms(boxes.cpu().numpy(), s3.numpy(), 0.5, seg3.numpy())
    keep, num, _ = torch.ops.tvmi.nms_step(boxes, s3.to(DEV), seg3.to(DEV), 0.5, 4, seg3.to(DEV), None, 4, 10)
    assert int(num) == len(want) and np.array_equal(keep[: int(num)].cpu().numpy(), want)
    torch.library.opcheck(torch.ops.tvmi.nms_step, args=(boxes[:200], scores[:200], seg3[:200].to(DEV), 0.5, 4, seg3[:200].to(DEV), None, 4, 10))


@pytest.mark.parametrize("C,P,dtype", [(256, 7, torch.float32), (256, 14, torch.float32), (256, 7, torch.bfloat16), (64, 7, torch.float32),
                                       (512, 7, torch.float16)])
def test_multiscale_roi_align_boxes_builds_its_own_rows(C, P, dtype):
    """tvmi::multiscale_roi_align_boxes (round 6: the [K,5] rows of convert_boxes_to_roi_format, ops/_utils.py:18-25, are written
    by the launch-order pre-pass of the call) == tvmi::boxes_to_rois + tvmi::multiscale_roi_align, bit for bit: output, the rows it
    returns, and the gradients of every level — with unequal and EMPTY per-image lists, on the fused route (a multiple of 256
    channels) and on the fallback (64 channels: rows built by tvmi_boxes_to_rois inside the C entry)."""
    g = gen(3100 + C + P)
    B = 3
    feats = [torch.randn(B, C, 200 // s, 304 // s, generator=g).to(DEV, dtype) for s in (1, 2, 4, 8)]
    boxes = [random_boxes(n, 1216, 800, 6, 500, g).to(DEV) for n in (700, 0, 1301)]
    scales = [0.25, 0.125, 0.0625, 0.03125]
    tail = (P, P, 2, False, 2, 5, 224.0, 4.0, 1e-6)
    f1 = [f.clone().requires_grad_(True) for f in feats]
    f2 = [f.clone().requires_grad_(True) for f in feats]
    out, rois = torch.ops.tvmi.multiscale_roi_align_boxes(f1, boxes, scales, *tail)
    want_rois = torch.ops.tvmi.boxes_to_rois(boxes)
    want = torch.ops.tvmi.multiscale_roi_align(f2, want_rois, scales, *tail)
    assert torch.equal(rois, want_rois) and rois.shape == (2001, 5)
    assert torch.equal(out, want)
    gr = torch.randn(out.shape, generator=g).to(DEV, dtype)
    out.backward(gr)
    want.backward(gr)
    for a, b in zip(f1, f2):
        assert torch.equal(a.grad, b.grad)
    # the module takes the one-op route for device boxes (and still equals the per-level oracle in the baseline-size tests)
    pool = vision_amd.MultiScaleRoIAlign(["0", "1", "2", "3"], P, 2)
    got = pool({str(i): f for i, f in enumerate(feats)}, boxes, [(800, 1216)] * B)
    assert torch.equal(got, want.detach())
    assert torch.ops.tvmi.multiscale_roi_align_boxes(feats, [b[:0] for b in boxes], scales, *tail)[0].shape == (0, C, P, P)


def test_nms_step_under_graph_capture_and_on_many_streams():
    """The hand-over words of tvmi::nms_step live in a block the library owns per (device, stream), re-armed by the kernel; a stream
    that is being CAPTURED gets the caller's workspace words + a memset node instead (no allocation inside a capture).  Both forms
    give the eager result: captured into a hipGraph and replayed 20 times, and eagerly on 8 streams at once."""
    g = gen(31337)
    n, S = 3000, 6
    boxes = random_boxes(n, 900, 700, 8, 200, g).to(DEV)
    scores = torch.rand(n, generator=g).to(DEV)
    seg = (torch.arange(n) * S // n)[torch.randperm(n, generator=g)].to(DEV)
    img = (seg % 3).contiguous()
    want = [t.clone() for t in torch.ops.tvmi.nms_step(boxes, scores, seg, 0.5, S, img, None, 3, 50)]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        torch.ops.tvmi.nms_step(boxes, scores, seg, 0.5, S, img, None, 3, 50)      # warm-up on the capture stream's pool
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = torch.ops.tvmi.nms_step(boxes, scores, seg, 0.5, S, img, None, 3, 50)
    for _ in range(20):
        for t in out:
            t.fill_(-5)   # (without this the comparison passed on memory the capture's allocations inherited: a memset NODE in
        graph.replay()    #  front of the kernel made every graph after the process's first replay correctly only once; the
        torch.cuda.synchronize()   # capture path zeroes its words with a kernel node now — nms_step_device.h, step_zero_words)
        for a, b in zip(out, want):
            assert torch.equal(a, b)
    # ... and a second graph in the same process
    graph2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph2):
        out2 = torch.ops.tvmi.nms_step(boxes, scores, seg, 0.5, S, img, None, 3, 50)
    for _ in range(5):
        for t in out2:
            t.fill_(-5)
        graph2.replay()
        torch.cuda.synchronize()
        for a, b in zip(out2, want):
            assert torch.equal(a, b)
    streams = [torch.cuda.Stream() for _ in range(8)]
    outs = []
    for _ in range(5):
        for st in streams:
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                outs.append(torch.ops.tvmi.nms_step(boxes, scores, seg, 0.5, S, img, None, 3, 50))
    torch.cuda.synchronize()
    for i, o in enumerate(outs):
        assert int(o[1]) == int(want[1]), (i, int(o[1]), int(want[1]))
        for a, b in zip(o, want):
            assert torch.equal(a, b), i


@pytest.mark.parametrize("C,P,dtype,n,S,B,kind", [(256, 7, torch.float32, 4000, 4, 4, "by_image"), (256, 7, torch.bfloat16, 3000, 40, 8, "dense"),
                                                  (512, 7, torch.float16, 70, 3, 3, "random"), (64, 7, torch.float32, 2500, 5, 5, "ties"),
                                                  (256, 14, torch.float32, 1024, 1, 1, "random")])
def test_one_launch_step_equals_the_two_ops(C, P, dtype, n, S, B, kind):
    """tvmi::roi_align_boxes_nms_step — the detector step as ONE launch: the NMS workgroups ride in front of the RoIAlign grid
    (roi_align_fwd_ms_dma_inl_step) — returns what tvmi::multiscale_roi_align_boxes and tvmi::nms_step return, bit for bit: on the
    carrying route (7x7 bins, a multiple of 256 channels), with the carrying switched off, and on shapes that cannot carry (64
    channels, 14x14 bins: the two entries one after the other).  Several calls in a row (the hand-over words are re-armed by every
    launch), through the module method bench.py uses, and captured into a hipGraph."""
    g = gen(5150 + C + P + n)
    feats = [torch.randn(3, C, 200 // s, 304 // s, generator=g).to(DEV, dtype) for s in (1, 2, 4, 8)]
    boxes = [random_boxes(m, 1216, 800, 6, 500, g).to(DEV) for m in (700, 0, 1301)]
    scales = [0.25, 0.125, 0.0625, 0.03125]
    tail = (P, P, 2, False, 2, 5, 224.0, 4.0, 1e-6)
    nb, ns, nseg, nimg, nlab = (t.to(DEV) for t in _step_case(n, S, B, g, kind))
    D = 100
    want_out, want_rois = torch.ops.tvmi.multiscale_roi_align_boxes(feats, boxes, scales, *tail)
    want_keep, want_num, want_pay = torch.ops.tvmi.nms_step(nb, ns, nseg, 0.5, S, nimg, nlab, B, D)
    k = int(want_num)

    def check(res):
        out, rois, keep, num, pay = res
        assert torch.equal(out, want_out) and torch.equal(rois, want_rois)
        assert int(num) == k, (int(num), k)
        assert torch.equal(keep[:k], want_keep[:k])
        assert torch.equal(pay, want_pay)

    args = (feats, boxes, scales, *tail, nb, ns, nseg, 0.5, S, nimg, nlab, B, D)
    for _ in range(3):
        check(torch.ops.tvmi.roi_align_boxes_nms_step(*args))
    torch.ops.tvmi.set_option("roi_align.carry_step", 0)
    try:
        check(torch.ops.tvmi.roi_align_boxes_nms_step(*args))
    finally:
        torch.ops.tvmi.set_option("roi_align.carry_step", 1)
    pool = vision_amd.MultiScaleRoIAlign(["0", "1", "2", "3"], P, 2)
    pooled, keep, num, pay = pool.forward_with_nms_step({str(i): f for i, f in enumerate(feats)}, boxes, [(800, 1216)] * 3, nb, ns, nseg, 0.5, S,
                                                        nimg, B, D, labels=nlab)
    assert torch.equal(pooled, want_out) and int(num) == k and torch.equal(keep[:k], want_keep[:k]) and torch.equal(pay, want_pay)
    # captured: the stream's hand-over block cannot be allocated inside a capture — the caller's words + a memset node instead
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        torch.ops.tvmi.roi_align_boxes_nms_step(*args)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    gph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gph):
        res = torch.ops.tvmi.roi_align_boxes_nms_step(*args)
    for _ in range(5):
        for t in res:
            t.fill_(-5)
        gph.replay()
        torch.cuda.synchronize()
        check(res)
