"""Repository contract: the product never touches the oracle; required artefacts exist."""
import os
import re

from helpers import ROOT


def _py_files(d):
    for base, _, files in os.walk(d):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                yield os.path.join(base, f)


def test_product_never_imports_the_oracle():
    pat = re.compile(r"^\s*(from|import)\s+oracle\b|oracle/|liboracle|libtv_ref_cpu", re.M)
    for path in _py_files(os.path.join(ROOT, "vision_amd")):
        text = open(path, errors="ignore").read()
        assert not pat.search(text), f"{path} references the oracle"


def test_required_files_exist():
    for rel in ("include/tvmi.h", "bench.py", "__graft_entry__.py", "DESIGN.md", "INTEGRATION.md", "oracle/README.md",
                "oracle/tvmi_oracle.c", "oracle/build_ref.py", "oracle/gen_golden.py", "tests/golden/nms.npz",
                "tests/golden/roi_ops.npz", "vision_amd/csrc/Makefile"):
        assert os.path.exists(os.path.join(ROOT, rel)), rel


def test_no_reference_sources_copied():
    # the oracle recipe compiles the reference where it lies; nothing named like its TUs lives here
    for base, _, files in os.walk(ROOT):
        if ".git" in base or "gpurun_out" in base:
            continue
        for f in files:
            assert not re.fullmatch(r"(nms|roi_align|roi_pool|ps_roi_align|ps_roi_pool|deform_conv2d|box_iou_rotated)_kernel\.(cpp|cu)", f), os.path.join(base, f)


def test_oracle_headers_say_test_infrastructure():
    for rel in ("oracle/tvmi_oracle.c", "oracle/oracle_impl.inc", "oracle/oracle.py", "oracle/build_ref.py",
                "oracle/gen_golden.py", "oracle/ref_compat_permute.h"):
        assert "TEST INFRASTRUCTURE ONLY" in open(os.path.join(ROOT, rel)).read(2000), rel
