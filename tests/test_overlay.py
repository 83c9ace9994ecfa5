"""Drop-in check against the reference's OWN python package: lay /root/reference/torchvision
over our operator library (symlinks only, nothing copied) and import it.
The reference's extension.py, _meta_registrations.py and _autograd_registrations.py must bind
to our schema definitions unchanged.  Runs only where /root/reference exists (not on the GPU box)."""
import os
import subprocess
import sys
import textwrap

import pytest

from helpers import ROOT

REF = "/root/reference/torchvision"


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")
def test_reference_python_package_binds_to_our_library(tmp_path):
    from vision_amd import integration

    overlay = integration.make_overlay(str(tmp_path / "overlay"), REF)
    code = textwrap.dedent(
        f"""
        import sys, torch
        sys.path.insert(0, {overlay!r}); sys.path.insert(0, {ROOT!r})
        import torchvision                      # the reference package, unmodified
        from torchvision import extension
        assert extension._has_ops(), "reference loader did not find _C/_C_stable"
        from oracle import oracle as O          # reference CPU kernels -> compute on CPU tensors
        torch.ops.load_library(O._REF)
        import torchvision.ops as ops
        b = torch.rand(50, 4) * 50; b[:, 2:] += b[:, :2]
        keep = ops.nms(b, torch.rand(50), 0.5)
        x = torch.rand(1, 4, 16, 16, requires_grad=True)
        y = ops.roi_align(x, [b[:5] / 4], 3, 1.0, 2)     # autograd formula from the reference package
        y.sum().backward()
        assert x.grad is not None and keep.dtype == torch.int64
        m = ops.MultiScaleRoIAlign(["0"], 3, 2)
        assert torch._C._dispatch_has_kernel_for_dispatch_key("torchvision::roi_align", "CUDA")
        print("OVERLAY_OK", torchvision.__file__)
        """
    )
    env = dict(os.environ, TVMI_NO_PY_REGISTRATIONS="1")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0 and "OVERLAY_OK" in out.stdout, out.stderr[-3000:]
