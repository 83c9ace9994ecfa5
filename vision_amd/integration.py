"""Laying the reference's python package over this operator library.

torchvision/extension.py:8-33 looks for `_C` and `_C_stable` shared objects NEXT TO the
package and `torch.ops.load_library`s them.  `make_overlay` builds a directory holding a
`torchvision/` whose entries are symlinks to an existing reference checkout / install plus
`_C.so` -> our `tvmi_torch.so` and `_C_stable.so` -> our `tvmi_torch_stable.so`; put it first on sys.path and
`import torchvision` — nothing of the reference is copied or modified.  Set
TVMI_NO_PY_REGISTRATIONS=1 if `vision_amd` is imported in the same process: the reference
package brings its own fake/autograd/autocast registrations for the same schemas.
"""
import os

from . import _loader


def make_overlay(dst: str, reference_pkg: str) -> str:
    """Create `<dst>/torchvision` (symlinks) and return `dst`."""
    if not os.path.isdir(reference_pkg) or not os.path.exists(os.path.join(reference_pkg, "extension.py")):
        raise FileNotFoundError(f"{reference_pkg} is not a torchvision package directory")
    for so in (_loader.KERNELS_SO, _loader.SHIM_SO, _loader.STABLE_SHIM_SO):
        if not os.path.exists(so):
            raise _loader.ExtensionMissing(f"{so} is missing; build the extension first")
    pkg = os.path.join(dst, "torchvision")
    os.makedirs(pkg, exist_ok=True)
    for name in os.listdir(reference_pkg):
        if name in ("__pycache__",) or name.startswith("_C"):
            continue
        link = os.path.join(pkg, name)
        if not os.path.lexists(link):
            os.symlink(os.path.join(reference_pkg, name), link)
    for name, target in (("_C.so", _loader.SHIM_SO), ("_C_stable.so", _loader.STABLE_SHIM_SO)):
        link = os.path.join(pkg, name)
        if os.path.lexists(link):
            os.remove(link)
        os.symlink(target, link)
    return dst
