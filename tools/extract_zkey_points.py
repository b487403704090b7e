#!/usr/bin/env python3
"""Extract the BN254 G1 points the reference holds as DATA into tests/golden/zkey_g1_points.json.

Source (present only in the build container; this script never runs on the GPU box):
  /root/reference/example-app/test-vectors/circom/multiplier2_final.zkey   (3 035 bytes, snarkjs Groth16 proving key)
It is the only place the reference tree holds curve POINTS rather than constants.  The container format (iden3 binfile: magic "zkey", version,
sections of (type u32, size u64); Groth16 header = section 2, IC = 3, A = 5, B1 = 6, C = 8, H = 9) stores a G1 point as 2 x 32 bytes of
little-endian **R = 2^256 Montgomery** words -- exactly what include/msm_hip.h calls MSM_FORM_MONT ("bit-identical to arkworks Fq.0") -- and the
point at infinity as 64 zero bytes.  The fixture keeps the words as they are (hex of the 64 bytes), the section each came from and an infinity
flag; this script also CHECKS, with Python integers, that the header's q is the BN254 base-field modulus and that every non-zero entry, after
multiplying by 2^-256, satisfies y^2 = x^3 + 3.  Data only: no text of the reference travels.
"""
import json
import os
import random
import struct
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bn254_py as ec  # independent pure-Python group law (the generator of tests/golden/msm_*.npz)

SRC = "/root/reference/example-app/test-vectors/circom/multiplier2_final.zkey"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "zkey_g1_points.json")
P = 21888242871839275222246405745257275088696311157297823662689037894645226208583
R_INV = pow(1 << 256, -1, P)
NAMES = {3: "IC", 5: "A", 6: "B1", 8: "C", 9: "H"}


def main():
    raw = open(SRC, "rb").read()
    assert raw[:4] == b"zkey"
    version, nsec = struct.unpack_from("<II", raw, 4)
    pos, sections = 12, {}
    for _ in range(nsec):
        typ, size = struct.unpack_from("<IQ", raw, pos)
        sections[typ] = raw[pos + 12:pos + 12 + size]
        pos += 12 + size
    assert pos == len(raw)
    assert struct.unpack_from("<I", sections[1], 0)[0] == 1, "not a Groth16 key"
    h = sections[2]
    n8q = struct.unpack_from("<I", h, 0)[0]
    q = int.from_bytes(h[4:4 + n8q], "little")
    o = 4 + n8q
    n8r = struct.unpack_from("<I", h, o)[0]
    r = int.from_bytes(h[o + 4:o + 4 + n8r], "little")
    o += 4 + n8r
    n_vars, n_public, domain = struct.unpack_from("<III", h, o)
    o += 12
    assert n8q == 32 and q == P, "header modulus is not the BN254 base field"
    g1 = 2 * n8q
    points = []

    def take(buf, off, section, index):
        b = buf[off:off + g1]
        assert len(b) == g1
        inf = not any(b)
        if not inf:
            x = int.from_bytes(b[:32], "little") * R_INV % P
            y = int.from_bytes(b[32:], "little") * R_INV % P
            assert int.from_bytes(b[:32], "little") < P and int.from_bytes(b[32:], "little") < P, "word not canonical"
            assert (y * y - x * x * x - 3) % P == 0, (section, index, "not on y^2 = x^3 + 3 after leaving the Montgomery domain")
        points.append({"section": section, "index": index, "infinity": bool(inf), "mont_le_hex": b.hex()})

    take(h, o, "alpha1", 0)
    take(h, o + g1, "beta1", 0)
    take(h, o + 2 * g1 + 2 * 2 * g1, "delta1", 0)  # alpha1, beta1, beta2 (G2), gamma2 (G2), delta1, delta2 (G2)
    assert len(h) == o + 3 * g1 + 3 * 2 * g1
    for typ in sorted(NAMES):
        buf = sections[typ]
        assert len(buf) % g1 == 0
        for i in range(len(buf) // g1):
            take(buf, i * g1, NAMES[typ], i)
    assert len(sections[3]) // g1 == n_public + 1 and len(sections[5]) // g1 == n_vars and len(sections[9]) // g1 == domain
    # known answers for the MSM tests: fixed scalars (seeded; a zero, a one, r - 1 and a full-width value among them) against the 19 entries in
    # file order, expected sum by the pure-Python big-integer group law -- affine, standard form
    rnd = random.Random(0xB2545A)
    scalars = [rnd.randrange(r) for _ in points]
    scalars[1], scalars[4], scalars[7], scalars[10] = 0, 1, r - 1, (1 << 253) + 12345
    affine = []
    for p_ in points:
        b = bytes.fromhex(p_["mont_le_hex"])
        affine.append(None if p_["infinity"] else (int.from_bytes(b[:32], "little") * R_INV % P, int.from_bytes(b[32:], "little") * R_INV % P))
    total = None
    for pt, k in zip(affine, scalars):
        total = ec.add(total, ec.mul(k, pt))
    assert total is not None and ec.is_on_curve(total)
    out = {
        "_source": "example-app/test-vectors/circom/multiplier2_final.zkey of the reference: G1 entries of the Groth16 header and of sections 3, 5, 6, 8, 9",
        "_format": "mont_le_hex = the 64 bytes as stored: x || y, each 32 bytes little-endian, R = 2^256 Montgomery (MSM_FORM_MONT); infinity = 64 zero bytes",
        "q_hex": hex(q), "r_hex": hex(r), "n_vars": n_vars, "n_public": n_public, "domain_size": domain,
        "points": points,
        "scalars_hex": [hex(k) for k in scalars],
        "expected_msm_affine_std_hex": [hex(total[0]), hex(total[1])],
        "expected_sum_of_points_affine_std_hex": [hex(c) for c in __import__("functools").reduce(ec.add, affine, None)],
    }
    json.dump(out, open(OUT, "w"), indent=1)
    nz = sum(1 for p_ in points if not p_["infinity"])
    print(f"{len(points)} G1 entries ({nz} points on the curve, {len(points) - nz} at infinity) -> {os.path.normpath(OUT)}")


if __name__ == "__main__":
    main()
