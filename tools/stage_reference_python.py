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


TESTS_ARCHIVE = os.path.join(ROOT, "_staged", "reference_tests.tar.gz")
# The reference's OWN tests of the hot path: test/test_ops.py (RoIOpTester, TestNMS, TestDeformConv,
# TestRotatedBoxIou, opcheck), test/test_models.py (test_detection_model + its expect pickles), and what they import.
TEST_FILES = ["test_ops.py", "test_models.py", "common_utils.py", "conftest.py", "_utils_internal.py", "optests_failures_dict.json", "assets/masks.tiff",
              "assets/encode_jpeg/grace_hopper_517x606.jpg"]
TEST_EXPECT = ("rcnn", "retinanet", "fcos", "ssd")


def stage_tests(verbose=True):
    """Archives the reference's own test files for the hot path (git-ignored, like the python package): they are run
    UNCHANGED against our operator library by tests/run_reference_tests.py."""
    src = os.path.join(REF, "test")
    if not os.path.isdir(src):
        return TESTS_ARCHIVE if os.path.exists(TESTS_ARCHIVE) else None
    os.makedirs(os.path.dirname(TESTS_ARCHIVE), exist_ok=True)
    names = list(TEST_FILES)
    names += sorted(os.path.join("expect", f) for f in os.listdir(os.path.join(src, "expect"))
                    if f.startswith("ModelTester.test_") and any(k in f for k in TEST_EXPECT))
    with tarfile.open(TESTS_ARCHIVE, "w:gz") as tar:
        for n in names:
            tar.add(os.path.join(src, n), arcname=os.path.join("reference_tests", n))
    if verbose:
        print(f"[stage] {len(names)} reference test files archived in {TESTS_ARCHIVE}")
    return TESTS_ARCHIVE


def reference_tests(scratch):
    """Directory holding the reference's test files: the live checkout, else the archive unpacked under `scratch`, else None."""
    live = os.path.join(REF, "test")
    if os.path.exists(os.path.join(live, "test_ops.py")):
        return live
    if not os.path.exists(TESTS_ARCHIVE):
        return None
    dst = os.path.join(scratch, "reference_tests")
    if not os.path.exists(os.path.join(dst, "test_ops.py")):
        with tarfile.open(TESTS_ARCHIVE, "r:gz") as tar:
            tar.extractall(scratch)
    return dst


if __name__ == "__main__":
    stage()
    stage_tests()
    sys.exit(0)
