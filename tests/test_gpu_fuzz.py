"""Randomised parity: tools/fuzz_parity.py draws image sizes, batch sizes, engine capacities and every DualTVL1 parameter
the engine supports, and compares flows and executed iteration counts with the oracle bit for bit."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [11, 12])
def test_random_sizes_batches_and_parameters_match_the_oracle(seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "20", str(seed)], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "20/20 cases identical" in r.stdout


def test_deepflow_random_sizes_and_batches_match_the_oracle():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_deepflow.py"), "8", "5"], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "8/8 cases identical" in r.stdout
