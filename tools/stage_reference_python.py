#!/usr/bin/env python3
"""Stages the reference's PYTHON package (`torchvision/**/*.py`; nothing from csrc, no data files) as ONE archive,
_staged/reference_python.tar.gz.

Why: the drop-in claim is "the unchanged reference python (torchvision.ops, models.detection) runs on our operator
library".  The GPU box has no /root/reference, so the package is archived into a git-ignored directory that travels
with the gpurun snapshot (same mechanism as the prebuilt oracle/_ref library); there it is unpacked into a scratch
directory and laid over our library with `vision_amd.integration.make_overlay` — tests/test_overlay.py (CUDA tensors)
and `bench.py --e2e` (config 5) use it.  The archive is never tracked (.gitignore) and nothing in vision_amd/ reads it:
it is the CALLER side of the boundary, neither part of the product nor of the oracle.
"""
import io
import os
import sys
import tarfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("TVMI_REFERENCE_ROOT", "/root/reference")
ARCHIVE = os.path.join(ROOT, "_staged", "reference_python.tar.gz")


def stage(verbose=True):
    """(Re)creates the archive where the reference checkout exists; returns its path or None."""
    src = os.path.join(REF, "torchvision")
    if not os.path.isdir(src):
        if verbose:
            print(f"[stage] {REF} not present; archive {'present' if os.path.exists(ARCHIVE) else 'absent'}")
        return ARCHIVE if os.path.exists(ARCHIVE) else None
    os.makedirs(os.path.dirname(ARCHIVE), exist_ok=True)
    n = 0
    with tarfile.open(ARCHIVE, "w:gz") as tar:
        for root, dirs, files in os.walk(src):
            dirs[:] = sorted(d for d in dirs if d not in ("csrc", "__pycache__"))
            for f in sorted(files):
                if f.endswith(".py"):
                    full = os.path.join(root, f)
                    tar.add(full, arcname=os.path.join("torchvision", os.path.relpath(full, src)))
                    n += 1
        # version.py is written by the reference's setup.py at build time; we do not run that
        vfile = os.path.join(REF, "version.txt")
        vtxt = open(vfile).read().strip() if os.path.exists(vfile) else "0.0.0"
        data = f"__version__ = '{vtxt}+tvmi.overlay'\ngit_version = 'unknown'\n".encode()
        info = tarfile.TarInfo("torchvision/version.py")
        info.size = len(data)
        tar.addfile(info, io.BytesIO(data))
    if verbose:
        print(f"[stage] {n} python files of the reference archived in {ARCHIVE}")
    return ARCHIVE


def reference_package(scratch):
    """A usable reference python package directory: the live checkout where it exists, else the archive unpacked
    under `scratch`, else None."""
    live = os.path.join(REF, "torchvision")
    if os.path.exists(os.path.join(live, "extension.py")):
        return live
    if not os.path.exists(ARCHIVE):
        return None
    dst = os.path.join(scratch, "reference_python")
    if not os.path.exists(os.path.join(dst, "torchvision", "extension.py")):
        os.makedirs(dst, exist_ok=True)
        with tarfile.open(ARCHIVE, "r:gz") as tar:
            tar.extractall(dst)
    return os.path.join(dst, "torchvision")


if __name__ == "__main__":
    stage()
    sys.exit(0)
