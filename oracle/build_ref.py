#!/usr/bin/env python3
"""oracle/build_ref.py — TEST INFRASTRUCTURE ONLY.

Compiles the REAL reference CPU kernels, from the sources where they lie under
/root/reference (never copied into this repo), into oracle/_ref/libtv_ref_cpu.so.

Only the `torchvision/csrc/ops/cpu/*_kernel.cpp` translation units are built: each of
them contains nothing but `STABLE_TORCH_LIBRARY_IMPL(torchvision, CPU, m)` blocks
(e.g. cpu/nms_kernel.cpp:134-136), so the library registers the reference's CPU-key
implementations under the `torchvision::` schemas that vision_amd's own
tvmi_torch.so defines.  With both loaded, `torch.ops.torchvision.X(cpu_tensor)` IS the
reference and `X(cuda_tensor)` is ours — the reference's own architecture.

The reference's build system (setup.py / cmake) is NOT run; this is plain g++ on 9 files (7 ops/cpu kernels + the two
ops/quantized/cpu kernels).
Outputs go only to oracle/_ref/ (git-ignored, but shipped to the GPU box by gpurun).
The GPU box has no /root/reference: there this script is a no-op that reports whether a
prebuilt library is present.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("TVMI_REFERENCE_ROOT", "/root/reference")
CSRC = os.path.join(REF, "torchvision", "csrc")
OUT_DIR = os.path.join(HERE, "_ref")
OUT = os.path.join(OUT_DIR, "libtv_ref_cpu.so")
KERNELS = ["nms", "roi_align", "roi_pool", "ps_roi_align", "ps_roi_pool", "deform_conv2d", "box_iou_rotated"]
# The two quantized TUs (CPU-only ops of the reference) also carry their schema DEFINITION (a STABLE_TORCH_LIBRARY_FRAGMENT
# block, quantized/cpu/qnms_kernel.cpp:148-150, qroi_align_kernel.cpp:234-237), which vision_amd's own library already provides:
# ref_compat_no_fragment.h (force-included, no reference source touched) turns that block into an unused function.
QUANTIZED = ["qnms", "qroi_align"]


def have_reference():
    return os.path.isdir(os.path.join(CSRC, "ops", "cpu"))


def build(force=False, verbose=True):
    if not have_reference():
        if verbose:
            print(f"[oracle/_ref] {REF} not present; prebuilt={os.path.exists(OUT)}")
        return OUT if os.path.exists(OUT) else None
    import torch

    srcs = [os.path.join(CSRC, "ops", "cpu", f"{k}_kernel.cpp") for k in KERNELS]
    qsrcs = [os.path.join(CSRC, "ops", "quantized", "cpu", f"{k}_kernel.cpp") for k in QUANTIZED]
    srcs = srcs + [q for q in qsrcs if os.path.exists(q)]
    newest = max(os.path.getmtime(p) for p in srcs + [os.path.join(HERE, "ref_compat_permute.h"),
                                                       os.path.join(HERE, "ref_compat_no_fragment.h"), __file__])
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= newest:
        return OUT
    os.makedirs(OUT_DIR, exist_ok=True)
    tdir = os.path.dirname(torch.__file__)
    objs = []
    procs = []
    for src in srcs:
        obj = os.path.join(OUT_DIR, os.path.basename(src).replace(".cpp", ".o"))
        cmd = [
            "g++", "-O2", "-std=c++17", "-fPIC", "-w",
            f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}",
            # torch 2.10 stable ABI (the reference asks for 2.14; see ref_compat_permute.h)
            "-DTORCH_TARGET_VERSION=0x020a000000000000",
            "-include", os.path.join(HERE, "ref_compat_permute.h"),
            *(["-include", os.path.join(HERE, "ref_compat_no_fragment.h")] if os.sep + "quantized" + os.sep in src else []),
            f"-I{CSRC}", f"-I{tdir}/include", f"-I{tdir}/include/torch/csrc/api/include",
            "-c", src, "-o", obj,
        ]
        procs.append((subprocess.Popen(cmd), cmd))
        objs.append(obj)
    for p, cmd in procs:
        if p.wait() != 0:
            raise RuntimeError("reference TU failed to compile: " + " ".join(cmd))
    link = ["g++", "-shared", "-o", OUT] + objs + [f"-L{tdir}/lib", "-ltorch", "-ltorch_cpu", "-lc10",
                                                   f"-Wl,-rpath,{tdir}/lib"]
    subprocess.check_call(link)
    for o in objs:
        os.remove(o)
    if verbose:
        print(f"[oracle/_ref] built {OUT}")
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
