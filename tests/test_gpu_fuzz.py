"""Seeded run of tools/fuzz_parity.py: random volumes, images, cameras and option combinations against the oracle."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_randomised_parity_sweep():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_parity.py"), "--cases", "120", "--slab-cases", "30", "--seed", "7"],
                       capture_output=True, text=True, timeout=170)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    assert "120 cases + 30 slab cases, 0 failures" in r.stdout
