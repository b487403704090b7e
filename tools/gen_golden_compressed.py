#!/usr/bin/env python3
"""tools/gen_golden_compressed.py -- golden vectors for the arkworks-0.4 compressed G1Affine image format
(the reference's instance files, mopro-msm/src/msm/utils/preprocess.rs:193-223).

Written with Python ints only (tools/bn254_py.py), independent of oracle/ and of the HIP code:
    image = x (standard form) | (y > p - y) << 255 | infinity << 254, 32 bytes little-endian.
Run ONCE in the build container; emits tests/golden/compressed_points.json (data only).
The format itself is restated from ark-serialize/ark-ec 0.4 (crates absent from /root/reference; no image file is
committed there), so these vectors pin the three implementations here against each other, not against arkworks.
"""
import json
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bn254_py as bn  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
P = bn.P


def compress(pt):
    if pt is bn.INF:
        return (1 << 254).to_bytes(32, "little")
    x, y = pt
    v = x | ((1 << 255) if y > P - y else 0)
    return v.to_bytes(32, "little")


def main():
    rng = random.Random(0xB254C0)
    pts = [bn.G, bn.neg(bn.G), bn.INF, bn.mul(2, bn.G), bn.mul(bn.R_ORDER - 2, bn.G)]
    for _ in range(59):
        pts.append(bn.mul(rng.randrange(1, bn.R_ORDER), bn.G))
    valid = []
    for pt in pts:
        assert pt is bn.INF or bn.is_on_curve(pt)
        valid.append({"image": compress(pt).hex(),
                      "inf": pt is bn.INF,
                      "x": "%064x" % (0 if pt is bn.INF else pt[0]),
                      "y": "%064x" % (0 if pt is bn.INF else pt[1])})
    # invalid images
    invalid = []
    x = 5
    while pow((x ** 3 + 3) % P, (P - 1) // 2, P) == 1:  # first x >= 5 whose x^3+3 is a non-residue
        x += 1
    invalid.append({"image": x.to_bytes(32, "little").hex(), "why": "x^3+3 is not a square"})
    invalid.append({"image": (x | (1 << 255)).to_bytes(32, "little").hex(), "why": "x^3+3 is not a square (negative flag)"})
    invalid.append({"image": P.to_bytes(32, "little").hex(), "why": "x == p"})
    invalid.append({"image": ((1 << 254) - 1).to_bytes(32, "little").hex(), "why": "x >= p (all ones below the flags)"})
    invalid.append({"image": (1 | (3 << 254)).to_bytes(32, "little").hex(), "why": "both flags set"})
    with open(os.path.join(OUT, "compressed_points.json"), "w") as f:
        json.dump({"generator": "tools/gen_golden_compressed.py", "seed": "0xB254C0",
                   "format": "ark-serialize 0.4 compressed SW affine: x LE | bit255 = y>p-y | bit254 = infinity",
                   "valid": valid, "invalid": invalid}, f, indent=1)
    print("wrote", len(valid), "valid and", len(invalid), "invalid images")


if __name__ == "__main__":
    main()
