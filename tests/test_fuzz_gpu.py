"""Randomised cross-checks (tools/fuzz_conv.py, tools/fuzz_chain.py): every eligible geometry of the fp16x3 convolution
engine returns the same bits and agrees with torch fp32; every fused bottleneck form returns the bits of the separate
convolutions — on random shapes, strides, residuals, activations and output formats; the wide halo-tile kernel (loader / compute
waves meeting at one barrier per tap) is launched repeatedly with other kernels in between (tools/fuzz_halo_wide.py)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("tool,cases,seed,env", [("fuzz_conv.py", 80, 11, {}), ("fuzz_chain.py", 36, 12, {}), ("fuzz_halo_wide.py", 60, 13, {}),
                                                 # the 256-row kernel's persistent form with 8..64 workgroups walking many tiles each
                                                 ("fuzz_conv.py", 60, 14, {"FCP_BIG_PERSIST": "1"})])
def test_randomised_bit_identity(tool, cases, seed, env):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), str(cases), str(seed)], capture_output=True, text=True,
                       timeout=600, cwd=ROOT, env={**os.environ, **env})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.strip().endswith(f"{cases} cases, 0 bad"), r.stdout[-2000:]
