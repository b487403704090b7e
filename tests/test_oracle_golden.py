"""CPU tests: pin the oracle (oracle/bn254_oracle.c) against
  (a) the literal constants the reference commits (tests/golden/reference_constants.json,
      extracted from SH/constants.metal, mont_params.rs:116-122, barrett_params.rs:25-28),
  (b) EFD known answers 2G, 3G, (r-1)G, rG,
  (c) the independent pure-Python MSM vectors tests/golden/msm_*.npz.
"""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden_cases, load_golden, load_srs_sets, load_zkey_points
from oracle import bn254_oracle as orc

P, R = orc.P, orc.R_ORDER
REFC = json.load(open(os.path.join(GOLDEN, "reference_constants.json")))


def w(v):
    return orc.int_to_words(v)


def test_constants_match_reference_literals():
    p, r1, r2, inv = orc.fq_constants()
    assert p == int(REFC["BN254_BASEFIELD_MODULUS"]["hex"], 16) == P
    assert r1 == int(REFC["BN254_ZERO_XR"]["hex"], 16) == (1 << 256) % P
    assert r2 == pow(1 << 256, 2, P)
    assert inv == (-pow(P, -1, 1 << 64)) % (1 << 64)
    assert inv & 0xFFFF == REFC["N0"] == REFC["N0_TEST"] == 25481
    assert int(REFC["RINV_DECIMAL"]) == pow(1 << 256, -1, P)
    assert int(REFC["BARRETT_MU_DECIMAL"]) == (1 << 508) // P == int(REFC["BARRETT_MU"]["hex"], 16)
    assert int(REFC["MONT_RADIX"]["hex"], 16) == 1 << 256


def test_to_mont_of_generator_matches_reference():
    # BN254_ONE_{X,Y,Z}R = Montgomery image of (1,2,1), constants.metal:229-282
    assert orc.words_to_int(orc.fq_to_mont(w(1))) == int(REFC["BN254_ONE_XR"]["hex"], 16)
    assert orc.words_to_int(orc.fq_to_mont(w(2))) == int(REFC["BN254_ONE_YR"]["hex"], 16)
    assert orc.words_to_int(orc.fq_from_mont(w(int(REFC["BN254_ONE_YR"]["hex"], 16)))) == 2


def test_field_ops_vs_python_ints():
    import random
    rng = random.Random(7)
    Rm = 1 << 256
    vals = [0, 1, 2, P - 1, P - 2, (Rm % P), rng.randrange(P)] + [rng.randrange(P) for _ in range(200)]
    for i in range(len(vals) - 1):
        a, b = vals[i], vals[i + 1]
        assert orc.words_to_int(orc.fq_add(w(a), w(b))) == (a + b) % P
        assert orc.words_to_int(orc.fq_sub(w(a), w(b))) == (a - b) % P
        assert orc.words_to_int(orc.fq_mont_mul(w(a), w(b))) == a * b * pow(Rm, -1, P) % P
        assert orc.words_to_int(orc.fq_to_mont(w(a))) == a * Rm % P
    a = vals[7]
    am = a * Rm % P
    assert orc.words_to_int(orc.fq_inv_mont(w(am))) == pow(a, -1, P) * Rm % P


def _gen_jac():
    return np.concatenate([orc.fq_to_mont(w(1)), orc.fq_to_mont(w(2)), orc.fq_to_mont(w(1))])


def test_curve_known_answers():
    g = _gen_jac()
    g2, inf = orc.g1_to_affine_std(orc.g1_dbl(g))
    assert inf == 0
    assert orc.words_to_int(g2[:8]) == 0x030644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD3
    assert orc.words_to_int(g2[8:]) == 0x15ED738C0E0A7C92E7845F96B2AE9C0A68A6A449E3538FC7FF3EBF7A5A18A2C4
    g3, _ = orc.g1_to_affine_std(orc.g1_add(orc.g1_dbl(g), g))
    assert orc.words_to_int(g3[:8]) == 0x0769BF9AC56BEA3FF40232BCB1B6BD159315D84715B8E679F2D355961915ABF0
    assert orc.words_to_int(g3[8:]) == 0x2AB799BEE0489429554FDB7C8D086475319E63B40B9C5B57CDF1FF3DD9FE2261
    # madd agrees with add; G+G through add() takes the doubling branch; P + (-P) = inf
    g_aff_m = np.concatenate([orc.fq_to_mont(w(1)), orc.fq_to_mont(w(2))])
    a3, _ = orc.g1_to_affine_std(orc.g1_madd(orc.g1_dbl(g), g_aff_m))
    assert (a3 == g3).all()
    d, _ = orc.g1_to_affine_std(orc.g1_add(g, g))
    assert (d == g2).all()
    dm, _ = orc.g1_to_affine_std(orc.g1_madd(g, g_aff_m))
    assert (dm == g2).all()
    gxy = np.concatenate([w(1), w(2)])
    m1, inf = orc.g1_to_affine_std(orc.g1_scalar_mul(gxy, w(R - 1)))
    assert inf == 0 and orc.words_to_int(m1[:8]) == 1 and orc.words_to_int(m1[8:]) == P - 2
    _, inf = orc.g1_to_affine_std(orc.g1_scalar_mul(gxy, w(R)))
    assert inf == 1
    _, inf = orc.g1_to_affine_std(orc.g1_add(orc.g1_scalar_mul(gxy, w(R - 1)), g))
    assert inf == 1
    # identity is (1,1,0) in Montgomery form: constants.metal:175-228
    z = orc.g1_scalar_mul(gxy, w(0))
    assert orc.words_to_int(z[16:]) == 0 and orc.words_to_int(z[:8]) == int(REFC["BN254_ZERO_XR"]["hex"], 16)


@pytest.mark.parametrize("name", golden_cases())
def test_msm_golden(name):
    g = load_golden(name)
    n = g["bases"].shape[0]
    algos = [("pippenger", lambda: orc.msm_pippenger(g["bases"], g["scalars"], orc.FORM_STD, g["inf"])),
             ("cuzk-ref-table", lambda: orc.msm_cuzk(g["bases"], g["scalars"], orc.FORM_STD, g["inf"]))]
    if n <= 1024:
        algos.append(("naive", lambda: orc.msm_naive(g["bases"], g["scalars"], orc.FORM_STD, g["inf"])))
    if n <= 256:
        for wb in (13, 15, 16):
            algos.append((f"cuzk-w{wb}", lambda wb=wb: orc.msm_cuzk(g["bases"], g["scalars"], orc.FORM_STD, g["inf"], wb)))
    for label, fn in algos:
        out, inf, _ = fn()
        assert inf == int(g["expected_inf"]), (name, label)
        assert (out == g["expected"]).all(), (name, label)


def test_mont_form_inputs_agree():
    g = load_golden("rand_n17")
    bm = np.stack([np.concatenate([orc.fq_to_mont(b[:8]), orc.fq_to_mont(b[8:])]) for b in g["bases"]])
    out, inf, jac = orc.msm_pippenger(bm, g["scalars"], orc.FORM_MONT, g["inf"])
    assert inf == 0 and (out == g["expected"]).all()
    aff, _ = orc.g1_to_affine_std(jac)
    assert (aff == out).all()


def test_reference_digit_semantics():
    """Appendix B of SURVEY.md: stored = digit + H, digit in [-H, H-1], sum digit_i * 2^(w i) == scalar."""
    g = load_golden("edge_carry_patterns")
    for wb in (8, 13, 15, 16):
        ch = orc.decompose_signed(g["scalars"], wb).astype(np.int64)
        H = 1 << (wb - 1)
        assert ch.min() >= 0 and ch.max() < 2 * H
        for i in range(g["scalars"].shape[0]):
            s = orc.words_to_int(g["scalars"][i])
            assert sum(int(ch[k, i] - H) << (wb * k) for k in range(ch.shape[0])) == s
        C = 2 * H
        col_ptr, val = orc.transpose(ch.astype(np.uint32), C)
        for k in range(ch.shape[0]):
            assert col_ptr[k, -1] == ch.shape[1]
            order = np.argsort(ch[k], kind="stable")
            assert (val[k] == order).all()


def test_synthetic_generator_closed_form():
    k = orc.gen_scalars(0xB2540001, 300, nonzero=True)
    s = orc.gen_scalars(0xB2540002, 300)
    bases = orc.gen_bases_from_logs(k, orc.FORM_MONT)
    out, inf, _ = orc.msm_pippenger(bases, s, orc.FORM_MONT)
    exp, einf = orc.closed_form_expected(k, s)
    assert inf == einf == 0 and (out == exp).all()
    # element i depends only on (seed, i)
    assert (orc.gen_scalars(0xB2540002, 10) == s[:10]).all()


def test_reference_zkey_points_pin_the_montgomery_word_format():
    """The one file of the reference that holds G1 POINTS (example-app/test-vectors/circom/multiplier2_final.zkey, extracted as data by
    tools/extract_zkey_points.py): 64-byte entries of little-endian R = 2^256 Montgomery words, infinity as zeros.  The oracle's
    fq_from_mont must put every non-zero entry on y^2 = x^3 + 3, its MSM over them in MONT form must equal the pure-Python known answer,
    and the same call on the standard-form images must give the same words -- which pins what include/msm_hip.h calls MSM_FORM_MONT
    ("bit-identical to arkworks Fq.0") against bytes the reference itself ships."""
    bases, inf, scalars, expected, d = load_zkey_points()
    assert int(d["q_hex"], 16) == P and int(d["r_hex"], 16) == R
    assert bases.shape == (19, 16) and int(inf.sum()) == 4
    std = np.zeros_like(bases)
    for i in range(bases.shape[0]):
        if inf[i]:
            assert not bases[i].any()
            continue
        x, y = orc.fq_from_mont(bases[i, :8]), orc.fq_from_mont(bases[i, 8:])
        xi, yi = orc.words_to_int(x), orc.words_to_int(y)
        assert xi < P and yi < P and (yi * yi - xi * xi * xi - 3) % P == 0, ("not on the curve after fq_from_mont", i)
        assert (orc.fq_to_mont(x) == bases[i, :8]).all() and (orc.fq_to_mont(y) == bases[i, 8:]).all()
        std[i, :8], std[i, 8:] = x, y
    for label, fn in (("pippenger/mont", lambda: orc.msm_pippenger(bases, scalars, orc.FORM_MONT, inf)),
                      ("naive/mont", lambda: orc.msm_naive(bases, scalars, orc.FORM_MONT, inf)),
                      ("cuzk/mont", lambda: orc.msm_cuzk(bases, scalars, orc.FORM_MONT, inf)),
                      ("pippenger/std", lambda: orc.msm_pippenger(std, scalars, orc.FORM_STD, inf))):
        out, is_inf, _ = fn()
        assert is_inf == 0 and (out == expected).all(), label
    ones = np.zeros_like(scalars)
    ones[:, 0] = 1
    out, is_inf, _ = orc.msm_naive(bases, ones, orc.FORM_MONT, inf)
    exp_sum = np.concatenate([np.frombuffer(int(v, 16).to_bytes(32, "little"), dtype="<u4") for v in d["expected_sum_of_points_affine_std_hex"]])
    assert is_inf == 0 and (out == exp_sum).all()


def test_reference_srs_files_hold_msm_known_answers():
    """The only MSM KNOWN ANSWERS in the reference tree: its halo2 KZG parameter files (example-app/ios/{plonk,gemini}_fibonacci_srs.bin, extracted as
    data by tools/extract_srs_points.py) hold the monomial basis g[j] = tau^j G AND the Lagrange basis g_lagrange[i] = L_i(tau) G of the same tau, so
        g[j] = sum_i omega^(i*j) * g_lagrange[i]        for every j < 2^k
    is an MSM whose inputs and expected output are both reference-held bytes (R = 2^256 Montgomery words), computed by halo2curves -- not by this
    repo, not by its Python generator.  The oracle must reproduce all 8 + 16 of them in MONT form, through every algorithm it has."""
    sets = load_srs_sets()
    assert [(f, k) for f, k, *_ in sets] == [("plonk_fibonacci_srs.bin", 3), ("gemini_fibonacci_srs.bin", 4)]
    for fname, k, omega, g, gl in sets:
        n = 1 << k
        assert pow(omega, n, R) == 1 and pow(omega, n // 2, R) == R - 1 and g.shape == (n, 16) and gl.shape == (n, 16)
        one = np.stack([orc.int_to_words(1)] * n)
        out, inf, _ = orc.msm_naive(gl, one, orc.FORM_MONT)
        gen = np.concatenate([orc.int_to_words(1), orc.int_to_words(2)])
        assert inf == 0 and (out == gen).all(), "sum of the Lagrange basis is the generator"
        for j in range(n):
            scalars = np.stack([orc.int_to_words(pow(omega, i * j, R)) for i in range(n)])
            exp = np.concatenate([orc.fq_from_mont(g[j, :8]), orc.fq_from_mont(g[j, 8:])])  # the reference-held g[j], leaving the Montgomery domain
            for label, fn in (("naive", orc.msm_naive), ("pippenger", orc.msm_pippenger), ("cuzk", orc.msm_cuzk)):
                out, inf, _ = fn(gl, scalars, orc.FORM_MONT)
                assert inf == 0 and (out == exp).all(), (fname, j, label)
