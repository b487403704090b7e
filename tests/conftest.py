import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gpu-acceleration_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The built libraries are git-ignored; build them in-tree when a fresh checkout lacks them (hipcc cross-compiles
    gfx950 without a GPU).  The product library is never replaced by anything else: no build => tests fail loudly."""
    import subprocess
    pkg = os.path.join(ROOT, "gpu-acceleration_amd")
    if not (os.path.exists(os.path.join(pkg, "libmsm_hip.so")) and os.path.exists(os.path.join(pkg, "libmsm_hip_hooks.so"))):
        subprocess.check_call(["make", "-s", "-j2", "-C", os.path.join(pkg, "csrc")])  # product + hooks build
    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle_bn254.so")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])


def golden_cases():
    import json
    with open(os.path.join(GOLDEN, "index.json")) as f:
        return [c["name"] for c in json.load(f)["cases"]]


def load_golden(name):
    import numpy as np
    return np.load(os.path.join(GOLDEN, f"msm_{name}.npz"))


def load_zkey_points():
    """tests/golden/zkey_g1_points.json (tools/extract_zkey_points.py): the G1 entries of the reference's own Groth16 proving key
    (example-app/test-vectors/circom/multiplier2_final.zkey) -- the only POINTS the reference tree holds -- as arkworks / MSM_FORM_MONT words.
    Returns (bases_mont [n,16] u32, inf [n] u8, scalars [n,8] u32, expected affine standard-form words [16] u32, the json)."""
    import json
    import numpy as np
    d = json.load(open(os.path.join(GOLDEN, "zkey_g1_points.json")))
    pts = d["points"]
    bases = np.stack([np.frombuffer(bytes.fromhex(p["mont_le_hex"]), dtype="<u4") for p in pts]).astype(np.uint32)
    inf = np.array([1 if p["infinity"] else 0 for p in pts], np.uint8)
    words = lambda v: np.frombuffer(int(v, 16).to_bytes(32, "little"), dtype="<u4").astype(np.uint32)
    scalars = np.stack([words(k) for k in d["scalars_hex"]])
    expected = np.concatenate([words(v) for v in d["expected_msm_affine_std_hex"]])
    return bases, inf, scalars, expected, d


def load_srs_sets():
    """tests/golden/srs_kzg_points.json (tools/extract_srs_points.py): the halo2 KZG parameter files the reference ships -- for each, (k, omega,
    g [2^k,16] u32, g_lagrange [2^k,16] u32), all points as arkworks / MSM_FORM_MONT words exactly as stored.  g[j] = sum_i omega^(i*j) g_lagrange[i]:
    MSM inputs AND expected outputs held by the reference, written by an implementation this repo shares nothing with."""
    import json
    import numpy as np
    d = json.load(open(os.path.join(GOLDEN, "srs_kzg_points.json")))
    out = []
    for st in d["sets"]:
        pts = lambda hexes: np.stack([np.frombuffer(bytes.fromhex(h), dtype="<u4") for h in hexes]).astype(np.uint32)
        out.append((st["file"], st["k"], int(st["omega_hex"], 16), pts(st["g_mont_le_hex"]), pts(st["g_lagrange_mont_le_hex"])))
    return out
