#!/usr/bin/env python3
"""Generate tests/golden/msm_*.npz -- BN254 G1 MSM known-answer vectors.

Why this exists: the reference (zkmopro/gpu-acceleration, mopro-msm) commits no
golden MSM vectors; its e2e tests compare against arkworks `G::msm` on run-time
random inputs (MM/metal_msm.rs:739-760, T/cuzk/e2e.rs:14-63) and arkworks /
Rust are absent from this image.  The expected value of an MSM is a *group
element*, so any correct big-integer evaluation, normalised to affine, is
bit-identical to arkworks' `into_affine()`.  This script evaluates it twice,
independently:
  (1) naive   sum_i s_i * P_i          (affine adds with modular inverses)
  (2) closed  (sum_i s_i * k_i mod r) * G   because every base is P_i = k_i * G
and refuses to write a file unless both agree.

Run only in the dev container (pure Python, ~2-3 min):  python tools/gen_golden.py
Word format = what the reference's packer emits (limbs_conversion.rs:311-378):
8 little-endian u32 words of the standard-form (non-Montgomery) integer.
"""
import json
import os
import random
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bn254_py as ec  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
rng = random.Random(0xB254)


def rand_scalar():
    return rng.randrange(0, ec.R_ORDER)


def make_case(name, ks, scalars, note, naive=True):
    """ks[i] = discrete log of base i (0 => point at infinity)."""
    n = len(ks)
    pts = [ec.mul(k, ec.G) if k % ec.R_ORDER else None for k in ks]
    closed = ec.mul(sum(s * k for s, k in zip(scalars, ks)) % ec.R_ORDER, ec.G)
    if naive:
        nv = ec.msm_naive(pts, scalars)
        assert nv == closed, f"{name}: naive != closed form"
    assert all(ec.is_on_curve(p) for p in pts)
    bases = np.zeros((n, 16), dtype=np.uint32)
    inf = np.zeros((n,), dtype=np.uint8)
    for i, p in enumerate(pts):
        if p is None:
            inf[i] = 1  # arkworks G1Affine::identity(): x = y = 0, infinity = true
        else:
            bases[i, :8] = ec.to_words(p[0])
            bases[i, 8:] = ec.to_words(p[1])
    sc = np.array([ec.to_words(s) for s in scalars], dtype=np.uint32).reshape(n, 8)
    exp = np.zeros((16,), dtype=np.uint32)
    exp_inf = np.uint8(1 if closed is None else 0)
    if closed is not None:
        exp[:8] = ec.to_words(closed[0])
        exp[8:] = ec.to_words(closed[1])
    np.savez(
        os.path.join(OUT, f"msm_{name}.npz"),
        bases=bases, inf=inf, scalars=sc, expected=exp, expected_inf=exp_inf,
    )
    print(f"  {name}: n={n} inf_result={int(exp_inf)} ({note})")
    return {"name": name, "n": n, "note": note, "naive_checked": bool(naive)}


def main():
    os.makedirs(OUT, exist_ok=True)
    index = []
    r = ec.R_ORDER
    # --- uniform random cases -------------------------------------------------
    for n in (1, 2, 3, 17, 256, 1024, 4096):
        ks = [rng.randrange(1, r) for _ in range(n)]
        ss = [rand_scalar() for _ in range(n)]
        index.append(make_case(f"rand_n{n}", ks, ss, "uniform bases k*G, uniform scalars in [0,r)"))
    # --- edge cases the reference does not handle but arkworks does (SURVEY §4 gaps)
    n = 64
    ks = [rng.randrange(1, r) for _ in range(n)]
    ss = [rand_scalar() for _ in range(n)]
    ks2 = list(ks)
    for i in (0, 5, 63):
        ks2[i] = 0
    index.append(make_case("edge_inf_bases", ks2, ss, "bases 0,5,63 are the point at infinity"))
    ss2 = list(ss)
    for i in (1, 2, 40):
        ss2[i] = 0
    index.append(make_case("edge_zero_scalars", ks, ss2, "scalars 1,2,40 are zero"))
    index.append(make_case("edge_all_zero_scalars", ks, [0] * n, "all scalars zero => infinity"))
    index.append(make_case("edge_scalar_r_minus_1", ks, [r - 1] * n, "every scalar = r-1 (=-1)"))
    index.append(make_case("edge_same_base", [ks[0]] * n, ss, "64 copies of one base (P+P inside every bucket)"))
    index.append(make_case("edge_same_base_same_scalar", [ks[0]] * n, [ss[0]] * n,
                           "64 x (same base, same scalar): doubling chain in one bucket"))
    kk = []
    s3 = []
    for i in range(n // 2):
        kk += [ks[i], r - ks[i]]
        s3 += [ss[i], ss[i]]
    index.append(make_case("edge_p_minus_p", kk, s3, "pairs (P,-P) with equal scalars => infinity"))
    index.append(make_case("edge_small_scalars", ks, [rng.randrange(0, 1 << 32) for _ in range(n)],
                           "scalars < 2^32 (upper windows empty)"))
    index.append(make_case("edge_scalar_one", ks, [1] * n, "all scalars 1 => plain sum of bases"))
    # signed-digit carry chains: every 16/15/13/8-bit window at the recode threshold
    pats = []
    for w in (8, 13, 15, 16):
        h = 1 << (w - 1)
        for v in (h - 1, h, h + 1, (1 << w) - 1):
            s = 0
            for i in range(0, 254, w):
                s |= v << i
            pats.append(s % r)
    while len(pats) < n:
        pats.append(rand_scalar())
    index.append(make_case("edge_carry_patterns", ks, pats[:n],
                           "scalars whose windows sit at H-1, H, H+1, 2H-1 for w=8,13,15,16"))
    index.append(make_case("edge_generator", [1, 1, 2], [1, r - 2, 5],
                           "G*1 + G*(r-2) + 2G*5 = 9G"))
    with open(os.path.join(OUT, "index.json"), "w") as f:
        json.dump({"generator": "tools/gen_golden.py", "seed": "0xB254", "cases": index}, f, indent=1)


if __name__ == "__main__":
    main()
