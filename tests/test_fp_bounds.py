"""Limb-range proof-by-assertion for the lazily reduced 9 x 29-bit device field (CPU test, no GPU).

fp_bn254.hpp / ec_bn254.hpp are __host__ __device__; tools/fp_bounds_check.cpp compiles them for the host with
-DFP_BOUNDS_CHECK, which turns every range assumption (pad >= subtrahend limb, no 32-bit limb overflow, no 64-bit
column overflow, value < 2^261) into an abort, then drives random chains, boundary operands and worst-case
representatives (X + 6p, Y + 4p, ZZ + p ...) through the group law and compares field results with the
independent 4 x 64-bit host arithmetic.
"""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_fp_limb_bounds_hold(tmp_path):
    exe = tmp_path / "fp_bounds_check"
    cmd = ["hipcc", "-O2", "-std=c++17", "-DFP_BOUNDS_CHECK", "-x", "hip", "--offload-arch=gfx950",
           os.path.join(ROOT, "tools", "fp_bounds_check.cpp"), "-o", str(exe)]
    subprocess.run(cmd, check=True, capture_output=True, timeout=600)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "no bound violated" in r.stdout
