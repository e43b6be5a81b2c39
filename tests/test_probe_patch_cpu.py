"""The experiment switches live outside the product kernels (tools/probes/experiment_switches_r04.patch); tools/ledger.sh
needs the patch to apply to the current sources."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("patch") is None, reason="patch(1) not installed")
def test_experiment_patch_applies_to_current_sources():
    with open(os.path.join(ROOT, "tools", "probes", "experiment_switches_r04.patch")) as f:
        r = subprocess.run(["patch", "-p1", "--dry-run", "-s"], stdin=f, cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.skipif(shutil.which("patch") is None, reason="patch(1) not installed")
def test_epilogue_ab_patch_applies_to_current_sources():
    with open(os.path.join(ROOT, "tools", "probes", "epilogue_serial_r05.patch")) as f:
        r = subprocess.run(["patch", "-p1", "--dry-run", "-s"], stdin=f, cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_product_kernels_carry_no_experiment_switches():
    import re
    bad = []
    csrc = os.path.join(ROOT, "face-crop-plus_amd", "csrc")
    for name in os.listdir(csrc):
        if name.endswith((".hip", ".h")):
            for i, line in enumerate(open(os.path.join(csrc, name)), 1):
                if re.search(r"FCP_\w*(ABLATE|PROBE|_ROT\b|REG_EPI|STAGES3|BURST\)|SPREAD\)|PROFILING)", line) and line.lstrip().startswith("#"):
                    bad.append(f"{name}:{i}: {line.strip()}")
    assert not bad, bad
