"""Opt-in auto-fuse for users of the UNCHANGED reference python.

`TVMI_AUTOFUSE=1` in the environment when the reference's `torchvision` package is imported over this library
(the overlay's `_C.so` is our `tvmi_torch.so`) swaps, at CLASS level, the methods whose python loops cost
the detectors their time —

    torchvision.ops.poolers.MultiScaleRoIAlign.forward                       (ops/poolers.py:289-321, loop :199-222)
    torchvision.models.detection.roi_heads.RoIHeads.postprocess_detections   (roi_heads.py:680-737)
    torchvision.models.detection.rpn.RegionProposalNetwork.filter_proposals  (rpn.py:242-297)
    torchvision.models.detection.retinanet.RetinaNet.postprocess_detections  (retinanet.py:509-571)
    torchvision.models.detection.transform.GeneralizedRCNNTransform.forward / .postprocess (transform.py:119-276)
    torchvision.ops.roi_align._roi_align                                     (ops/roi_align.py:276-281: the python detour
        under torch.use_deterministic_algorithms(True) — our backward IS deterministic, so CUDA tensors stay on the op)

— for the method factories of `vision_amd.integration` (the ones `fuse_detection_model` binds to a single model).  Every
replacement calls the reference's own method whenever the fused path does not apply (CPU tensors, training / targets,
tracing or scripting, fixed_size transforms, a RetinaNet subclass with another coder).  Nothing of the reference is copied
or edited: the swap happens in memory, after the reference module has executed.

How it gets triggered without touching the reference: the static initialiser of `tvmi_torch.so` (torch_shim.cpp) — the
library torchvision/extension.py:8-33 loads — sees TVMI_AUTOFUSE and executes THIS file (stdlib imports only: the library is
still inside dlopen at that point, `vision_amd` itself is imported lazily at the first call of a swapped method) as the
module `vision_amd.autofuse` and calls `install()`, which installs a `sys.meta_path` hook that patches the five modules right after they are executed (and patches the ones that
are imported already).  `vision_amd.autofuse.install()` is the same switch for code that prefers to call it.
torch.jit.script of a model takes the source of the class methods: script the model BEFORE switching this on (or not at all).
"""
import importlib.abc
import importlib.util
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# module -> [(class name — None for a module-level function —, method name, factory in vision_amd.integration)]
_TARGETS = {
    "torchvision.ops.roi_align": [(None, "_roi_align", "make_deterministic_roi_align")],
    "torchvision.ops.poolers": [("MultiScaleRoIAlign", "forward", "make_pool_forward")],
    "torchvision.models.detection.roi_heads": [("RoIHeads", "postprocess_detections", "make_postprocess_detections")],
    "torchvision.models.detection.rpn": [("RegionProposalNetwork", "filter_proposals", "make_filter_proposals")],
    "torchvision.models.detection.retinanet": [("RetinaNet", "postprocess_detections", "make_retinanet_postprocess")],
    "torchvision.models.detection.transform": [("GeneralizedRCNNTransform", "forward", "make_transform_forward"),
                                               ("GeneralizedRCNNTransform", "postprocess", "make_transform_postprocess")],
}
_patched = []          # (class, method name, original) for uninstall()
_state = {"finder": None}


def _lazy_method(orig, factory_name, module):
    built = {}

    def method(*args, **kwargs):
        fn = built.get("fn")
        if fn is None:
            os.environ.setdefault("TVMI_NO_PY_REGISTRATIONS", "1")     # the reference package registered the torchvision:: fakes
            if _ROOT not in sys.path:
                sys.path.append(_ROOT)
            from vision_amd import integration

            factory = getattr(integration, factory_name)
            if factory_name == "make_transform_forward":
                fn = factory(orig, getattr(module, "ImageList", None))
            else:
                fn = factory(orig)
            built["fn"] = fn
        return fn(*args, **kwargs)

    method.__name__ = getattr(orig, "__name__", "method")
    method.__qualname__ = getattr(orig, "__qualname__", method.__name__)
    method.__doc__ = getattr(orig, "__doc__", None)
    method.__wrapped__ = orig
    method._tvmi_autofused = True
    return method


def _patch_module(module):
    for cls_name, meth, factory_name in _TARGETS.get(module.__name__, ()):
        cls = module if cls_name is None else getattr(module, cls_name, None)
        orig = None if cls is None else cls.__dict__.get(meth)
        if orig is None or getattr(orig, "_tvmi_autofused", False):
            continue
        setattr(cls, meth, _lazy_method(orig, factory_name, module))
        _patched.append((cls, meth, orig))


class _PatchingLoader(importlib.abc.Loader):
    def __init__(self, inner):
        self._inner = inner

    def create_module(self, spec):
        return self._inner.create_module(spec)

    def exec_module(self, module):
        self._inner.exec_module(module)
        _patch_module(module)

    def __getattr__(self, name):            # get_source / get_code / get_filename / is_package: inspect and torch.jit read them
        return getattr(self._inner, name)


class _Finder(importlib.abc.MetaPathFinder):
    def __init__(self):
        self._busy = set()

    def find_spec(self, name, path=None, target=None):
        if name not in _TARGETS or name in self._busy:
            return None
        self._busy.add(name)
        try:
            spec = importlib.util.find_spec(name)
        except (ImportError, ValueError):
            spec = None
        finally:
            self._busy.discard(name)
        if spec is None or spec.loader is None:
            return None
        spec.loader = _PatchingLoader(spec.loader)
        return spec


def install():
    """Switch the class-level swaps on (idempotent): modules already imported are patched now, the others when they are."""
    if _state["finder"] is None:
        _state["finder"] = _Finder()
        sys.meta_path.insert(0, _state["finder"])
    for name in _TARGETS:
        mod = sys.modules.get(name)
        if mod is not None and getattr(getattr(mod, "__spec__", None), "_initializing", False) is False:
            _patch_module(mod)
    return True


def uninstall():
    """Put the reference's methods back and remove the import hook."""
    while _patched:
        cls, meth, orig = _patched.pop()
        setattr(cls, meth, orig)
    if _state["finder"] is not None:
        try:
            sys.meta_path.remove(_state["finder"])
        except ValueError:
            pass
        _state["finder"] = None


def installed() -> bool:
    return _state["finder"] is not None
