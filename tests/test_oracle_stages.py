"""CPU test of the oracle's stage mirrors (oracle_bucket_sums / oracle_bit_sums) against the committed golden vectors: the
weighted bucket sums and the bit sums must rebuild the golden MSM -- so the stage-level GPU tests compare against pinned code."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import bn254_oracle as orc


def _digits(scalars, c, W):
    H = 1 << (c - 1)
    out = np.zeros((W, scalars.shape[0]), np.int64)
    for i, s in enumerate(scalars):
        v, carry = orc.words_to_int(s), 0
        for w in range(W):
            d = ((v >> (c * w)) & ((1 << c) - 1)) + carry
            carry = 0
            if d > H:
                d, carry = d - 2 * H, 1
            out[w, i] = d
    return out


def _scale(j, k):
    """k * P for a Jacobian point by double-and-add on the oracle's group operations"""
    acc = None
    for bit in bin(k)[2:]:
        if acc is not None:
            acc = orc.g1_dbl(acc)
        if bit == "1":
            acc = j.copy() if acc is None else orc.g1_add(acc, j)
    return acc


@pytest.mark.parametrize("name,c", [("rand_n256", 5), ("edge_inf_bases", 4), ("edge_p_minus_p", 6), ("rand_n17", 7)])
def test_bucket_and_bit_sums_rebuild_the_golden_msm(name, c):
    g = load_golden(name)
    W, nb = 254 // c + 1, 1 << (c - 1)
    kb = c - 1
    d = _digits(g["scalars"], c, W)
    buckets = orc.bucket_sums(g["bases"], d, nb, orc.FORM_STD, g["inf"])
    q = orc.bit_sums(buckets, W, nb)
    ident = np.zeros(24, np.uint32)
    total_b, total_q = ident.copy(), ident.copy()
    for w in range(W - 1, -1, -1):
        for _ in range(c):
            total_b, total_q = orc.g1_dbl(total_b), orc.g1_dbl(total_q)
        for b in range(nb):  # sum_b (b+1) * B[w][b]
            aff, inf = orc.g1_to_affine_std(buckets[w * nb + b])
            if not inf:
                total_b = orc.g1_add(total_b, _scale(buckets[w * nb + b], b + 1))
        sw = q[w, kb].copy()  # Q_all + sum_u 2^u Q_u
        for u in range(kb):
            aff, inf = orc.g1_to_affine_std(q[w, u])
            if not inf:
                sw = orc.g1_add(sw, _scale(q[w, u], 1 << u))
        total_q = orc.g1_add(total_q, sw)
    for tot in (total_b, total_q):
        aff, inf = orc.g1_to_affine_std(tot)
        assert inf == int(g["expected_inf"]) and (aff == g["expected"]).all()
