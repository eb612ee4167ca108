"""A short randomised parity run (tests/fuzz_runner.py: random SPD patterns, orderings,
not-positive-definite injections, plan flags, 1-3 right-hand sides) against the oracle."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [3, 11])
def test_randomised_parity(seed):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_runner.py"), "40", str(seed)],
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "40 cases, 0 failures" in out.stdout, out.stdout[-2000:]


@pytest.mark.gpu
def test_randomised_parity_wave_tile_kernel_everywhere():
    """The same run with every dense update through k_update3 (CHOLMOD_HIP_UPD3_MIN_TILES=1): ragged tiles,
    every contraction length, assign-mode blocks."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_runner.py"), "40", "7"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=dict(os.environ, CHOLMOD_HIP_UPD3_MIN_TILES="1"))
    assert out.returncode == 0, out.stdout + out.stderr
    assert "40 cases, 0 failures" in out.stdout, out.stdout[-2000:]


def test_randomised_parity_cpu_path():
    """The same run through the product's CPU path (Common->useGPU = 0), complex cases included."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_runner.py"), "16", "5"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=dict(os.environ, FUZZ_USE_GPU="0"))
    assert out.returncode == 0, out.stdout + out.stderr
    assert "16 cases, 0 failures" in out.stdout, out.stdout[-2000:]
