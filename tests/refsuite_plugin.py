"""pytest plugin for tests/run_reference_tests.py — TEST INFRASTRUCTURE.

Loaded with `-p refsuite_plugin` into the pytest process that runs the REFERENCE's own test files (test/test_ops.py,
test/test_models.py, unchanged) against the reference's own python package laid over our operator library
(vision_amd.integration.make_overlay).  It does two things the reference's build would have done:

  * registers the reference's CPU kernels (oracle/_ref/libtv_ref_cpu.so, compiled unmodified by oracle/build_ref.py) for
    the CPU dispatch key, so that the `cpu` halves of `cpu_and_cuda()` parametrisations and the CPU side of every
    CPU-vs-CUDA comparison in those tests (e.g. TestNMS.test_nms_gpu, test_ops.py:943-961) run the reference itself;
  * records which shared objects of ours the process mapped, so the log shows the CUDA halves ran on the HIP kernels.
"""
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    import torch
    import torchvision  # the overlay: reference python + our _C / _C_stable

    ref = os.path.join(ROOT, "oracle", "_ref", "libtv_ref_cpu.so")
    if not os.path.exists(ref):
        raise pytest.UsageError(f"{ref} missing: build it with oracle/build_ref.py where /root/reference exists")
    torch.ops.load_library(ref)
    config._tvmi_loaded = sorted({line.split()[-1] for line in open("/proc/self/maps")
                                  if "tvmi" in line or "libtv_ref_cpu" in line})


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    for line in _header(config):
        terminalreporter.write_line(line)


def _header(config):
    import torch
    import torchvision

    return [f"refsuite: torchvision {torchvision.__version__} from {os.path.dirname(torchvision.__file__)}",
            f"refsuite: torch {torch.__version__}, cuda available: {torch.cuda.is_available()}"
            + (f" ({torch.cuda.get_device_name(0)})" if torch.cuda.is_available() else ""),
            "refsuite: mapped " + ", ".join(os.path.basename(p) for p in getattr(config, "_tvmi_loaded", []))]


@pytest.fixture
def mocker():
    """pytest-mock is not in this image; test_models.py:69-75 only needs `mocker.patch(target, ...)` undone at teardown."""
    from unittest import mock

    class _Mocker:
        def __init__(self):
            self._patches = []

        def patch(self, target, *a, **k):
            p = mock.patch(target, *a, **k)
            self._patches.append(p)
            return p.start()

    m = _Mocker()
    yield m
    for p in reversed(m._patches):
        p.stop()
