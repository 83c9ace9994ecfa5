"""Locates and loads the three in-tree native libraries of vision_amd.

    _lib/libtvmi_kernels.so    hand-written gfx950 kernels behind the C ABI of include/tvmi.h
    _lib/tvmi_torch.so         dispatcher glue: the `torchvision::` schemas + the classic-ABI CUDA-key kernels
                               (role of `_C` in torchvision/extension.py:8-33)
    _lib/tvmi_torch_stable.so  stable-ABI glue: the CUDA-key kernels of `nms` and `box_iou_rotated`
                               (role of `_C_stable`; the same two ops the reference keeps on the stable ABI)

There is deliberately NO python / eager fallback: if the extension is missing the import
fails loudly (`TVMI_ALLOW_MISSING=1` only lets pure-host helpers import for docs/tooling).
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "_lib")
KERNELS_SO = os.path.join(LIB_DIR, "libtvmi_kernels.so")
SHIM_SO = os.path.join(LIB_DIR, "tvmi_torch.so")
STABLE_SHIM_SO = os.path.join(LIB_DIR, "tvmi_torch_stable.so")

_state = {"loaded": False, "kernels": None}


class ExtensionMissing(ImportError):
    pass


def is_loaded() -> bool:
    return _state["loaded"]


def load():
    """Load libtvmi_kernels.so, tvmi_torch.so and tvmi_torch_stable.so (idempotent)."""
    if _state["loaded"]:
        return
    for path in (KERNELS_SO, SHIM_SO, STABLE_SHIM_SO):
        if not os.path.exists(path):
            raise ExtensionMissing(
                f"vision_amd native extension not built: {path} is missing. "
                "Run `python -c 'import __graft_entry__ as g; g.build()'` (or `make -C vision_amd/csrc`)."
            )
    # torch is imported above, so its libamdhip64.so.7 is already mapped; the kernels
    # library binds to that single HIP runtime (same SONAME) instead of a second copy.
    _state["kernels"] = ctypes.CDLL(KERNELS_SO, mode=ctypes.RTLD_GLOBAL)
    torch.ops.load_library(SHIM_SO)         # defines the schemas: first
    torch.ops.load_library(STABLE_SHIM_SO)
    _state["loaded"] = True


def kernels() -> ctypes.CDLL:
    """ctypes handle on the C ABI (used by tests to check the exported symbols)."""
    load()
    return _state["kernels"]


def has_ops() -> bool:
    return _state["loaded"] and hasattr(torch.ops.torchvision, "nms")


def assert_has_ops():
    if not has_ops():
        raise RuntimeError(
            "vision_amd: the gfx950 operator library is not loaded; there is no eager fallback. "
            "Build it with __graft_entry__.build()."
        )
