"""Runs tools/race_screen.py: the hand-synchronised kernels (LDS rings, counted waits, in-place epilogues) at the full
BASELINE layer sizes, repeated, bitwise against the simpler kernel of the same contraction / the first run."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_race_screen_full_size_layers():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "race_screen.py"), "12"], capture_output=True, text=True, timeout=600)
    print(r.stdout[-3000:])
    assert r.returncode == 0 and "RACE SCREEN CLEAN" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
