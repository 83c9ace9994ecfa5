#!/usr/bin/env python3
"""Runs the REFERENCE's own test files for the hot path, unchanged, on our operator library — TEST INFRASTRUCTURE.

    python tests/run_reference_tests.py [--log profiles/r06_reference_test_ops.log] [--suite ops|models|all] [pytest args]

What runs: /root/reference/test/test_ops.py (RoIOpTester :111-277 incl. fp64 gradcheck and TorchScript, TestNMS :874-1048,
TestDeformConv :1060-1331, TestRotatedBoxIou :1841-2124, opcheck :761-795,1051-1057, the box utilities) and
test/test_models.py -k detection (:784-884, with the reference's expect pickles), collected by the reference's own
conftest.py.  The test files and the python package come from the git-ignored archives tools/stage_reference_python.py
makes where /root/reference exists (they travel to the GPU box with the gpurun snapshot; nothing of them is tracked).
`torchvision` is the reference's python over `_C.so -> tvmi_torch.so`, `_C_stable.so -> tvmi_torch_stable.so`
(vision_amd.integration.make_overlay); the CPU dispatch key is served by the reference's own CPU kernels
(oracle/_ref, see tests/refsuite_plugin.py) — so every `cuda` parametrisation exercises the HIP kernels and every `cpu` one the
reference.  Deselected by name, with the reason (nothing else is filtered):

  * `mps` parametrisations                — no MPS device (the reference's conftest skips them itself).
  * tests that need torch.compile / inductor are run as they are; failures are listed in the log, not hidden.
"""
import argparse
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.stage_reference_python import ARCHIVE, TESTS_ARCHIVE, stage, stage_tests  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log", default=None)
    ap.add_argument("--suite", default="all", choices=("ops", "models", "all"))
    ap.add_argument("--scratch", default=None)
    ap.add_argument("--diag", action="store_true", help="run tests/refsuite_diag.py inside the overlay instead of pytest")
    args, extra = ap.parse_known_args()

    stage(verbose=False)
    stage_tests(verbose=False)
    if not (os.path.exists(ARCHIVE) and os.path.exists(TESTS_ARCHIVE)):
        print("reference archives not staged (tools/stage_reference_python.py needs /root/reference)")
        return 2
    scratch = args.scratch or tempfile.mkdtemp(prefix="tvmi_refsuite_")
    import tarfile

    for arc, dst in ((ARCHIVE, os.path.join(scratch, "reference_python")), (TESTS_ARCHIVE, scratch)):
        os.makedirs(dst, exist_ok=True)
        with tarfile.open(arc, "r:gz") as tar:
            tar.extractall(dst)
    from vision_amd import integration

    overlay = integration.make_overlay(os.path.join(scratch, "overlay"), os.path.join(scratch, "reference_python", "torchvision"))
    tdir = os.path.join(scratch, "reference_tests")
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([overlay, os.path.join(ROOT, "tests"), env.get("PYTHONPATH", "")])
    env.pop("TVMI_AUTOFUSE", None)
    if args.diag:
        return subprocess.call([sys.executable, os.path.join(ROOT, "tests", "refsuite_diag.py"), *extra], cwd=tdir, env=env)
    rc = 0
    log = open(args.log, "w") if args.log else None
    runs = []
    if args.suite in ("ops", "all"):
        runs.append(["test_ops.py"])
    if args.suite in ("models", "all"):
        runs.append(["test_models.py", "-k", "detection"])
    for run in runs:
        cmd = [sys.executable, "-m", "pytest", "-p", "refsuite_plugin", "-p", "no:cacheprovider", "-rfEs", "-q",
               "--tb=short", *run, *extra]
        head = "$ (cd <scratch>/reference_tests) " + " ".join(cmd[1:]) + "\n"
        print(head, end="", flush=True)
        p = subprocess.Popen(cmd, cwd=tdir, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if log:
            log.write(head)
        for line in p.stdout:
            line = line.replace(scratch, "<scratch>")
            sys.stdout.write(line)
            if log:
                log.write(line)
        rc |= p.wait()
    if log:
        log.close()
    return rc


if __name__ == "__main__":
    sys.exit(main())
