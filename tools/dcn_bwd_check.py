"""deform_conv2d backward: every route of tvmi_deform_conv2d_backward (matrix-core kernels, direct kernels, 16-bit, fp64)
against the reference CPU kernels (oracle/_ref) on small problems — error statistics per gradient, nothing asserted — and
the config-4 timings of the fused backward (the round-3 library-GEMM route was removed in round 5).
python tools/dcn_bwd_check.py out.json"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, vision_amd
from oracle import oracle as O
dev = "cuda"; tv = torch.ops.torchvision
have_ref = O.load_reference()
NAMES = ("grad_input", "grad_weight", "grad_offset", "grad_mask", "grad_bias")
CFGS = [
    dict(tag="mfma og1 64->96", B=2, C=64, OC=96, H=20, W=24, k=(3, 3), groups=1, og=1, stride=(1, 1), pad=(1, 1), dil=(1, 1), mask=True),
    dict(tag="mfma og2 (32/og)", B=2, C=64, OC=96, H=20, W=24, k=(3, 3), groups=1, og=2, stride=(1, 1), pad=(1, 1), dil=(1, 1), mask=True),
    dict(tag="mfma tails 48->200 s2 d2 nomask", B=1, C=48, OC=200, H=13, W=17, k=(3, 3), groups=1, og=1, stride=(2, 1), pad=(1, 2), dil=(1, 2), mask=False),
    dict(tag="mfma groups2 64->64 k(1,3)", B=3, C=64, OC=64, H=11, W=9, k=(1, 3), groups=2, og=2, stride=(1, 1), pad=(0, 1), dil=(1, 1), mask=True),
    dict(tag="mfma 320->272 (2 channel chunks, 2 oc chunks)", B=1, C=320, OC=272, H=9, W=10, k=(3, 3), groups=1, og=1, stride=(1, 1), pad=(1, 1), dil=(1, 1), mask=True),
    dict(tag="direct og3 (12/og)", B=3, C=36, OC=40, H=11, W=9, k=(1, 3), groups=2, og=3, stride=(1, 1), pad=(0, 1), dil=(1, 1), mask=True),
    dict(tag="direct depthwise", B=2, C=8, OC=8, H=9, W=9, k=(3, 3), groups=8, og=1, stride=(1, 1), pad=(1, 1), dil=(1, 1), mask=False),
    dict(tag="direct reference-test cfg", B=2, C=6, OC=2, H=5, W=4, k=(3, 2), groups=2, og=3, stride=(2, 1), pad=(1, 0), dil=(2, 1), mask=True),
    dict(tag="mfma small offsets 64->64 (window only)", B=2, C=64, OC=64, H=30, W=40, k=(3, 3), groups=1, og=1, stride=(1, 1), pad=(1, 1), dil=(1, 1), mask=True, off_scale=0.5),
    dict(tag="mfma zero offsets (y = -1 rows)", B=1, C=32, OC=32, H=8, W=8, k=(3, 3), groups=1, og=1, stride=(1, 1), pad=(1, 1), dil=(1, 1), mask=True, zero_off=True),
]
def make(cfg, dtype, seed=31):
    g = torch.Generator().manual_seed(seed)
    kh, kw = cfg["k"]
    oh = (cfg["H"] + 2 * cfg["pad"][0] - (cfg["dil"][0] * (kh - 1) + 1)) // cfg["stride"][0] + 1
    ow = (cfg["W"] + 2 * cfg["pad"][1] - (cfg["dil"][1] * (kw - 1) + 1)) // cfg["stride"][1] + 1
    x = torch.randn(cfg["B"], cfg["C"], cfg["H"], cfg["W"], generator=g)
    w = torch.randn(cfg["OC"], cfg["C"] // cfg["groups"], kh, kw, generator=g) * 0.1
    off = torch.randn(cfg["B"], 2 * cfg["og"] * kh * kw, oh, ow, generator=g) * cfg.get("off_scale", 2)
    if cfg.get("zero_off"): off = torch.zeros_like(off)
    m = torch.rand(cfg["B"], cfg["og"] * kh * kw, oh, ow, generator=g)
    b = torch.randn(cfg["OC"], generator=g)
    gr = torch.randn(cfg["B"], cfg["OC"], oh, ow, generator=g)
    args = (*cfg["stride"], *cfg["pad"], *cfg["dil"], cfg["groups"], cfg["og"], cfg["mask"])
    return [t.to(dtype) for t in (gr, x, w, off, m, b)], args
out = {}
def check(tag, cfg, dtype, mfma=1):
    ts, args = make(cfg, dtype)
    torch.ops.tvmi.set_option("dcn.bwd_mfma", mfma)
    try:
        got = tv._deform_conv2d_backward(*[t.to(dev) for t in ts], *args)
        torch.cuda.synchronize()
    except Exception as e:  # noqa
        print(tag, "FAILED", repr(e)[:300], flush=True); out[tag] = {"error": repr(e)[:300]}; return
    finally:
        torch.ops.tvmi.set_option("dcn.bwd_mfma", 1)
    ref = tv._deform_conv2d_backward(*[t.double() if dtype == torch.float64 else t.float() for t in ts], *args)
    row = {}
    for nm, a, r in zip(NAMES, got, ref):
        a = a.double().cpu().numpy(); r = r.double().numpy()
        scale = max(1.0, float(np.abs(r).max()))
        err = np.abs(a - r)
        row[nm] = {"max_err_over_scale": float(err.max() / scale) if err.size else 0.0, "scale": scale,
                   "n_bad_1e-3": int((err > 1e-3 * scale).sum()), "n": int(err.size)}
    out[tag] = row
    print(tag, {k: (round(v["max_err_over_scale"], 7), v["n_bad_1e-3"]) for k, v in row.items()}, flush=True)
if have_ref:
    for cfg in CFGS:
        check(cfg["tag"] + " fp32", cfg, torch.float32)
    for cfg in CFGS[:5]:
        torch.ops.tvmi.set_option("dcn.bwd_owner", 0)
        check(cfg["tag"] + " fp32 window kernel (LDS atomics)", cfg, torch.float32)
        torch.ops.tvmi.set_option("dcn.bwd_window", 0)
        check(cfg["tag"] + " fp32 global atomics only", cfg, torch.float32)
        torch.ops.tvmi.set_option("dcn.bwd_window", 1)
        torch.ops.tvmi.set_option("dcn.bwd_owner", 1)
    check(CFGS[0]["tag"] + " fp32 forced direct", CFGS[0], torch.float32, mfma=0)
    check(CFGS[1]["tag"] + " fp32 forced direct", CFGS[1], torch.float32, mfma=0)
    check(CFGS[0]["tag"] + " bf16", CFGS[0], torch.bfloat16)
    check(CFGS[1]["tag"] + " fp16", CFGS[1], torch.float16)
    check(CFGS[6]["tag"] + " bf16", CFGS[6], torch.bfloat16)
    check(CFGS[7]["tag"] + " fp64", CFGS[7], torch.float64)
    check(CFGS[8]["tag"] + " bf16", CFGS[8], torch.bfloat16)
    check(CFGS[0]["tag"] + " fp64", CFGS[0], torch.float64)
else:
    print("no reference CPU kernels (oracle/_ref): values not checked")

# ---- config 4 timings
def tm(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): fn()
        b.record(); torch.cuda.synchronize(); best = min(best, a.elapsed_time(b) / n)
    return round(best, 4)
g = torch.Generator().manual_seed(0)
B, C, H, W, OC = 2, 256, 100, 136, 256
x = torch.randn(B, C, H, W, generator=g); off = torch.randn(B, 18, H, W, generator=g); m = torch.rand(B, 9, H, W, generator=g)
b = torch.randn(OC, generator=g); gr = torch.randn(B, OC, H, W, generator=g)
for groups in (1, 256):
    w = torch.randn(OC, C // groups, 3, 3, generator=g) * (0.01 if groups == 1 else 0.2)
    for dt in (torch.float32, torch.bfloat16):
        ts = [t.to(dev, dt) for t in (gr, x, w, off, m, b)]
        for route, opts in (("fused", {}), ("fused_window_lds_atomics", {"dcn.bwd_owner": 0}), ("fused_global_atomics", {"dcn.bwd_owner": 0, "dcn.bwd_window": 0}),
                            ("direct", {"dcn.bwd_mfma": 0})):
            if route in ("direct", "fused_global_atomics", "fused_window_lds_atomics") and groups != 1: continue
            if route == "direct" and dt != torch.float32: continue
            for k, v in opts.items(): torch.ops.tvmi.set_option(k, v)
            key = f"c4 backward g={groups} {str(dt).split('.')[-1]} {route}"
            try:
                out[key] = {"ms": tm(lambda: tv._deform_conv2d_backward(*ts, 1, 1, 1, 1, 1, 1, groups, 1, True))}
            except Exception as e:  # noqa
                out[key] = {"error": repr(e)[:300]}
            print(key, out[key], flush=True)
            torch.ops.tvmi.set_option("dcn.bwd_mfma", 1); torch.ops.tvmi.set_option("dcn.bwd_window", 1); torch.ops.tvmi.set_option("dcn.bwd_owner", 1)
if len(sys.argv) > 1: json.dump(out, open(sys.argv[1], "w"), indent=1)
