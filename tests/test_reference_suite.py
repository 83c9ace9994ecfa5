"""The reference's OWN test/test_ops.py, unchanged, against the HIP kernels (VERDICT r05 missing 1): every
`cuda`-parametrised case of RoIOpTester (:111-277, incl. fp64 gradcheck and TorchScript), TestNMS (:874-1048), TestDeformConv
(:1060-1331), TestRotatedBoxIou (:1841-2124) and the opcheck families must pass.  The test file comes from the git-ignored
archive tools/stage_reference_python.py makes (it travels to the GPU box with the snapshot); tests/run_reference_tests.py lays the
reference's python over our libraries and serves the CPU key with the reference's own CPU kernels (oracle/_ref).
The full logs (cpu + cuda halves, and test_models.py -k detection) are profiles/r06_reference_test_{ops,models}.log."""
import os
import re
import subprocess
import sys

import pytest

from helpers import ROOT


@pytest.mark.gpu
def test_reference_test_ops_cuda_cases_pass(tmp_path):
    sys.path.insert(0, ROOT)
    from tools.stage_reference_python import ARCHIVE, TESTS_ARCHIVE, stage, stage_tests

    stage(verbose=False)
    stage_tests(verbose=False)
    if not (os.path.exists(ARCHIVE) and os.path.exists(TESTS_ARCHIVE)):
        pytest.skip("reference archives not staged")
    log = tmp_path / "ref_ops.log"
    cmd = [sys.executable, os.path.join(ROOT, "tests", "run_reference_tests.py"), "--suite", "ops", "--scratch", str(tmp_path / "s"),
           "--log", str(log), "-m", "needs_cuda"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    text = log.read_text() if log.exists() else out.stdout + out.stderr
    tail = text[-3000:]
    m = re.search(r"(\d+) passed", text)
    assert out.returncode == 0 and m and " failed" not in text and " error" not in text.split("short test summary")[-1], tail
    assert int(m.group(1)) >= 400, tail          # 460 cuda-parametrised cases are collected; a few are skipped by the reference itself
    assert "tvmi_torch.so" in text and "libtvmi_kernels.so" in text, tail   # the plugin's header: our libraries were mapped
