#!/usr/bin/env python3
"""Extract the KZG parameter files the reference ships as DATA into tests/golden/srs_kzg_points.json -- the only MSM KNOWN ANSWERS the reference tree holds.

Sources (present only in the build container; this script never runs on the GPU box):
  /root/reference/example-app/ios/{plonk,gemini}_fibonacci_srs.bin   (halo2 `ParamsKZG` in raw-bytes form: u32 k, then 2^k points g[j] = tau^j * G,
  then 2^k points g_lagrange[i] = L_i(tau) * G, then G2 data; every G1 point = x || y, 32 bytes each, little-endian R = 2^256 Montgomery words = MSM_FORM_MONT)
The two bases are tied by the inverse DFT over the 2^k-th roots of unity of Fr:   g[j] = sum_i omega^(i*j) * g_lagrange[i]   for every j -- an MSM whose INPUTS
(g_lagrange, the powers of omega) and EXPECTED OUTPUT (g[j]) are both in the file, computed by the halo2 / halo2curves code that wrote it: an implementation this
repo shares nothing with.  The script checks, with the independent pure-Python group law, that every point is on y^2 = x^3 + 3 after leaving the Montgomery domain, that
sum_i g_lagrange[i] = g[0] = the generator, finds the primitive root omega for which g[1] holds, verifies ALL j, and writes points + omega as data (hex of the stored bytes).
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bn254_py as ec

SRC = "/root/reference/example-app/ios"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "srs_kzg_points.json")
P, R = ec.P, ec.R_ORDER
R_INV = pow(1 << 256, -1, P)


def main():
    sets = []
    # (hyperplonk_fibonacci_srs.bin beside them has the same header and 32 points on the curve, but its second block is not the DFT image of
    # the first -- a multilinear basis, presumably -- so it carries no relation this script can state: left out)
    for name in ("plonk_fibonacci_srs.bin", "gemini_fibonacci_srs.bin"):
        b = open(os.path.join(SRC, name), "rb").read()
        k = int.from_bytes(b[:4], "little")
        n = 1 << k
        assert 1 <= k <= 8 and len(b) >= 4 + 2 * n * 64

        def pt(rec):
            raw = b[4 + 64 * rec:4 + 64 * rec + 64]
            xm, ym = int.from_bytes(raw[:32], "little"), int.from_bytes(raw[32:], "little")
            assert xm < P and ym < P, "word not canonical"
            p_ = (xm * R_INV % P, ym * R_INV % P)
            assert ec.is_on_curve(p_), (name, rec)
            return raw.hex(), p_

        g = [pt(j) for j in range(n)]
        gl = [pt(n + i) for i in range(n)]
        assert g[0][1] == (1, 2), "g[0] is not the generator"
        s = None
        for _, p_ in gl:
            s = ec.add(s, p_)
        assert s == (1, 2), "sum of the Lagrange basis is not the generator"
        w0 = pow(5, (R - 1) // n, R)  # 5 generates Fr*: every primitive 2^k-th root is an odd power of this one
        omega = None
        for e in range(1, n, 2):
            w = pow(w0, e, R)
            acc = None
            for i in range(n):
                acc = ec.add(acc, ec.mul(pow(w, i, R), gl[i][1]))
            if acc == g[1][1]:
                omega = w
                break
        assert omega is not None, "no primitive root reproduces g[1]"
        for j in range(n):  # ... and then every j must hold
            acc = None
            for i in range(n):
                acc = ec.add(acc, ec.mul(pow(omega, i * j, R), gl[i][1]))
            assert acc == g[j][1], (name, j)
        sets.append({"file": name, "k": k, "omega_hex": hex(omega), "g_mont_le_hex": [h for h, _ in g], "g_lagrange_mont_le_hex": [h for h, _ in gl]})
        print(f"{name}: k = {k}, {2 * n} G1 points, omega = {hex(omega)[:18]}..., g[j] = sum_i omega^(ij) g_lagrange[i] holds for all {n} j")
    out = {"_source": "example-app/ios/*_fibonacci_srs.bin of the reference: halo2 ParamsKZG, raw-bytes form",
           "_format": "*_mont_le_hex = the 64 bytes as stored: x || y, each 32 bytes little-endian, R = 2^256 Montgomery (MSM_FORM_MONT)",
           "_relation": "g[j] = sum_i omega^(i*j) * g_lagrange[i] (mod r in the exponents), for every j < 2^k: inputs and expected outputs are all from the file",
           "sets": sets}
    json.dump(out, open(OUT, "w"), indent=1)
    print("->", os.path.normpath(OUT))


if __name__ == "__main__":
    main()
