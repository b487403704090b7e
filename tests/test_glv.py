"""GLV split (csrc/glv_bn254.hpp) checked on the CPU: the device function is __host__ __device__, tools/glv_check.cpp runs it on
20000 scalars, Python integers verify k1 + lambda*k2 = k (mod r), |k_j| < 2^126, and the constants themselves."""
import json
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
P = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47


def _consts():
    return json.load(open(os.path.join(ROOT, "tests", "golden", "glv_constants.json")))


def test_glv_constants_are_consistent():
    c = _consts()
    lam, beta = int(c["lambda"], 16), int(c["beta"], 16)
    assert (lam * lam + lam + 1) % R == 0 and (beta * beta + beta + 1) % P == 0 and lam != 1 and beta != 1
    a1, b1, a2, b2 = (int(c[k]) for k in ("a1", "b1", "a2", "b2"))
    assert (a1 + b1 * lam) % R == 0 and (a2 + b2 * lam) % R == 0 and abs(a1 * b2 - a2 * b1) == R
    # the header carries the same numbers
    hdr = open(os.path.join(ROOT, "gpu-acceleration_amd", "csrc", "glv_bn254.hpp")).read()
    words = lambda v, n: ", ".join("0x%08xu" % ((v >> (32 * i)) & 0xFFFFFFFF) for i in range(n))
    assert words(beta, 8) in hdr and words(abs(b1), 4) in hdr and words(abs(a2), 4) in hdr and words(a1, 2) in hdr
    assert words(int(c["g1"], 16), 4) in hdr and words(int(c["g2"], 16), 6) in hdr and c["quotient_shift"] == 288
    # the 126-bit bound of the halves: quotients within 1/2 + 2^-35 of the exact ones (g = round(2^288 |b| / r), k < 2^254)
    from fractions import Fraction
    eps = Fraction(1, 2) + Fraction(1, 1 << 35)
    assert eps * (abs(a1) + abs(a2)) < 7 << 123 and eps * (abs(b1) + abs(b2)) < 7 << 123  # msm_planner.hpp glv_top_digit_bits relies on it


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_glv_split_identity_and_bound(tmp_path):
    exe = tmp_path / "glv_check"
    subprocess.run(["hipcc", "-O2", "-std=c++17", "-x", "hip", "--offload-arch=gfx950", os.path.join(ROOT, "tools", "glv_check.cpp"),
                    "-o", str(exe)], check=True, capture_output=True, timeout=600)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True, timeout=600).stdout.splitlines()
    lam = int(_consts()["lambda"], 16)
    assert len(out) == 20000
    worst = 0
    for line in out:
        ok, k, n1, k1, n2, k2 = line.split()
        k = int(k, 16)
        k1 = int(k1, 16) * (-1 if n1 == "1" else 1)
        k2 = int(k2, 16) * (-1 if n2 == "1" else 1)
        assert ok == "1" and (k1 + lam * k2 - k) % R == 0
        worst = max(worst, abs(k1), abs(k2))
    assert worst < 7 << 123
