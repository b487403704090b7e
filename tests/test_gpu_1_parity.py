"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the CPU oracle
(oracle/bn254_oracle.c), the committed golden vectors, and size-independent properties at
BASELINE.json's full sizes.  Bit-exact everywhere: this is integer work."""
import numpy as np
import pytest

import mopro_msm_hip as mh
from mopro_msm_hip import testhooks as th
from conftest import golden_cases, load_golden, load_srs_sets, load_zkey_points
from oracle import bn254_oracle as orc

pytestmark = pytest.mark.gpu
P, R = orc.P, orc.R_ORDER


@pytest.fixture(scope="module")
def ctx():
    """the PRODUCT library (libmsm_hip.so): every end-to-end parity test runs through it"""
    c = mh.MsmContext()
    yield c
    c.close()


@pytest.fixture(scope="module")
def hk():
    """the hooks build (libmsm_hip_hooks.so): device-math unit tests, the synthetic-instance generator, calibration"""
    c = th.HooksContext()
    yield c
    c.close()


def rand_fp(rng, n):
    vals = [0, 1, 2, P - 1, P - 2, (1 << 256) % P, (1 << 255) % P] + [int(rng.integers(0, 2**63)) for _ in range(4)]
    while len(vals) < n:
        vals.append(int.from_bytes(rng.bytes(32), "little") % P)
    return np.stack([orc.int_to_words(v) for v in vals[:n]])


# ---- device math library (counterpart of T/field, T/bigint, T/mont_backend tests) -------------
def test_fp_ops_match_oracle(hk):
    rng = np.random.default_rng(1)
    n = 600
    a, b = rand_fp(rng, n), rand_fp(rng, n)[::-1].copy()
    table = [(0, orc.fq_add), (1, orc.fq_sub), (2, orc.fq_mont_mul)]
    for op, ref in table:
        out = hk.test_fp_op(op, a, b)
        exp = np.stack([ref(a[i], b[i]) for i in range(n)])
        assert (out == exp).all(), f"fp op {op}"
    for op, ref in [(3, orc.fq_to_mont), (4, orc.fq_from_mont)]:
        out = hk.test_fp_op(op, a)
        exp = np.stack([ref(a[i]) for i in range(n)])
        assert (out == exp).all(), f"fp op {op}"
    m = 40
    out = hk.test_fp_op(5, a[:m])
    exp = np.stack([orc.fq_inv_mont(a[i]) for i in range(m)])
    assert (out == exp).all()
    # a * a^-1 == R (Montgomery one), skipping a == 0
    prod = hk.test_fp_op(2, a[1:m], out[1:m])
    assert all(orc.words_to_int(p) == (1 << 256) % P for p in prod)


def _jac_points(n, seed):
    k = orc.gen_scalars(seed, n, nonzero=True)
    g = np.concatenate([orc.int_to_words(1), orc.int_to_words(2)])
    return np.stack([orc.g1_scalar_mul(g, k[i]) for i in range(n)])


def _affine_of(jacs):
    return [orc.g1_to_affine_std(j) for j in jacs]


def _neg(j):
    o = j.copy()
    o[8:16] = orc.int_to_words((P - orc.words_to_int(j[8:16])) % P)
    return o


def _same_affine(x, y):
    ax, ai = orc.g1_to_affine_std(x)
    bx, bi = orc.g1_to_affine_std(y)
    return ai == bi and (ax == bx).all()


def test_g1_ops_match_oracle(hk):
    """madd / add / dbl incl. 0+Q, P+0, 0+0, P+P, P+P with different Z, P+(-P)
    (T/curve/jacobian_add_2007_b1.rs:121-180 cases, plus the ones the reference gets wrong)."""
    n = 40
    a = _jac_points(n, 11)
    b = _jac_points(n, 12)
    ident = np.zeros(24, np.uint32)
    ident[:8] = orc.fq_to_mont(orc.int_to_words(1))
    ident[8:16] = ident[:8]
    a[0] = ident
    b[1] = ident
    a[2] = ident
    b[2] = ident
    b[3] = a[3]
    b[4] = orc.g1_add(orc.g1_dbl(a[4]), _neg(a[4]))  # same point as a4, different Z
    b[5] = _neg(a[5])
    b[6] = _neg(orc.g1_add(orc.g1_dbl(a[6]), _neg(a[6])))  # -a6 with a different Z
    assert _same_affine(a[4], b[4]) and not (a[4] == b[4]).all()
    out = hk.test_g1_op(1, a, b)
    for i in range(n):
        assert _same_affine(out[i], orc.g1_add(a[i], b[i])), ("add", i)
    assert orc.g1_to_affine_std(out[5])[1] == 1 and orc.g1_to_affine_std(out[6])[1] == 1
    out = hk.test_g1_op(2, a)
    for i in range(n):
        assert _same_affine(out[i], orc.g1_dbl(a[i])), ("dbl", i)
    # mixed add: q affine Montgomery (identity cannot be an affine operand: use a fresh point there)
    spare = _jac_points(2, 13)
    baff = []
    for i in range(n):
        src = b[i] if i not in (1, 2) else spare[i - 1]
        xy, inf = orc.g1_to_affine_std(src)
        assert inf == 0
        baff.append(np.concatenate([orc.fq_to_mont(xy[:8]), orc.fq_to_mont(xy[8:])]))
    baff = np.stack(baff)
    out = hk.test_g1_op(0, a, baff)
    for i in range(n):
        assert _same_affine(out[i], orc.g1_madd(a[i], baff[i])), ("madd", i)
    assert orc.g1_to_affine_std(out[5])[1] == 1  # P + (-P)
    assert _same_affine(out[3], orc.g1_dbl(a[3])) and _same_affine(out[4], orc.g1_dbl(a[4]))
    # the same mixed additions with b's arkworks words gathered as they are (round 5: k_accumulate_pieces<.., M256> -- 32 * W unreduced as the
    # multiplier, the digit's sign applied to S2), both signs: a + b and a - b
    out = hk.test_g1_op(4, a, baff)
    for i in range(n):
        assert _same_affine(out[i], orc.g1_madd(a[i], baff[i])), ("madd m256", i)
    assert orc.g1_to_affine_std(out[5])[1] == 1 and _same_affine(out[3], orc.g1_dbl(a[3])) and _same_affine(out[4], orc.g1_dbl(a[4]))
    nbaff = baff.copy()
    for i in range(n):
        nbaff[i, 8:] = orc.fq_to_mont(orc.int_to_words((P - orc.words_to_int(orc.fq_from_mont(baff[i, 8:]))) % P))
    out = hk.test_g1_op(5, a, baff)
    for i in range(n):
        assert _same_affine(out[i], orc.g1_madd(a[i], nbaff[i])), ("madd m256 negated", i)
    out = hk.test_g1_op(5, a, nbaff)  # a - (-b) = a + b: the doubling and cancelling cases through the negated path
    for i in range(n):
        assert _same_affine(out[i], orc.g1_madd(a[i], baff[i])), ("madd m256 doubly negated", i)
    assert orc.g1_to_affine_std(out[5])[1] == 1 and _same_affine(out[3], orc.g1_dbl(a[3]))


def test_wide_add_matches_scalar_add_and_oracle(hk):
    """ec_wide.hpp: the 8-lane addition used by the reduction trees == the scalar complete addition == the oracle,
    on random pairs and on every special case (identity operands, P + P, P + P with another Z, P + (-P)),
    at group counts that leave partial wavefronts."""
    for n in (1, 7, 8, 9, 203):
        a = _jac_points(n, 21 + n)
        b = _jac_points(n, 22 + n)
        ident = np.zeros(24, np.uint32)
        ident[:8] = orc.fq_to_mont(orc.int_to_words(1))
        ident[8:16] = ident[:8]
        if n >= 7:
            a[0] = ident
            b[1] = ident
            a[2] = ident
            b[2] = ident
            b[3] = a[3]
            b[4] = orc.g1_add(orc.g1_dbl(a[4]), _neg(a[4]))
            b[5] = _neg(a[5])
            b[6] = _neg(orc.g1_add(orc.g1_dbl(a[6]), _neg(a[6])))
        wide = hk.test_g1_op(3, a, b)
        scalar = hk.test_g1_op(1, a, b)
        for i in range(n):
            assert _same_affine(wide[i], orc.g1_add(a[i], b[i])), ("wide add", n, i)
            assert _same_affine(wide[i], scalar[i]), ("wide vs scalar", n, i)
        if n >= 7:
            assert orc.g1_to_affine_std(wide[5])[1] == 1 and orc.g1_to_affine_std(wide[6])[1] == 1 and orc.g1_to_affine_std(wide[2])[1] == 1


def test_calibration_reports_plausible_multiplier_rates(hk):
    """msm_calibrate (the live peaks bench.py prices k_accumulate against): MI355X sustains ~3.9e13 v_mad_u64_u32/s and
    ~1.6e11 field multiplications/s; a field multiplication is 171 multiplier instructions plus bookkeeping."""
    mad, fpm = hk.calibrate()
    assert 5e12 < mad < 2e14 and 2e10 < fpm < 1e12
    assert 171 < mad / fpm < 400


def test_signed_digits_reconstruct_scalar(hk):
    g = load_golden("edge_carry_patterns")
    sc = np.concatenate([g["scalars"], orc.gen_scalars(5, 200), np.stack([orc.int_to_words(v) for v in (0, 1, R - 1, R - 2, (1 << 253) + 12345)])])
    for wb in (4, 8, 13, 15, 16, 17):
        d = hk.test_decompose(sc, wb).astype(object)
        H = 1 << (wb - 1)
        assert d.shape[0] == 254 // wb + 1
        assert d.max() <= H and d.min() >= -(H - 1)
        for i in range(sc.shape[0]):
            assert sum(int(d[k, i]) << (wb * k) for k in range(d.shape[0])) == orc.words_to_int(sc[i]), (wb, i)


# ---- end to end (counterpart of T/cuzk/e2e.rs:14-63, metal_msm.rs:739-760) ---------------------
@pytest.mark.parametrize("name", golden_cases())
def test_msm_golden_default_plan(ctx, name):
    g = load_golden(name)
    r = ctx.msm(g["bases"], g["scalars"], mh.FORM_STD, g["inf"])
    assert r.is_infinity == bool(g["expected_inf"]), name
    assert (r.affine_std == g["expected"]).all(), name
    aff, inf = orc.g1_to_affine_std(r.jacobian_mont)
    assert inf == int(g["expected_inf"]) and (aff == g["expected"]).all()


@pytest.mark.parametrize("wb,flags", [(8, 0), (13, 0), (15, 0), (16, 0), (16, mh.FLAG_UNSIGNED_DIGITS), (5, 0), (11, mh.FLAG_UNSIGNED_DIGITS),
                                      (0, mh.FLAG_NO_GLV), (8, mh.FLAG_NO_GLV), (13, mh.FLAG_NO_GLV), (16, mh.FLAG_NO_GLV),
                                      (16, mh.FLAG_UNSIGNED_DIGITS | mh.FLAG_NO_GLV), (11, mh.FLAG_UNSIGNED_DIGITS | mh.FLAG_NO_GLV)])
def test_msm_golden_window_overrides(wb, flags):
    with mh.MsmContext(window_bits=wb, flags=flags) as c:
        for name in golden_cases():
            g = load_golden(name)
            if g["bases"].shape[0] > 1024 and wb < 8:
                continue
            r = c.msm(g["bases"], g["scalars"], mh.FORM_STD, g["inf"])
            assert r.is_infinity == bool(g["expected_inf"]), (name, wb)
            assert (r.affine_std == g["expected"]).all(), (name, wb)


def test_mont_form_and_resident_bases(ctx, hk):
    g = load_golden("rand_n1024")
    bm = np.concatenate([hk.test_fp_op(3, g["bases"][:, :8]), hk.test_fp_op(3, g["bases"][:, 8:])], axis=1)
    r = ctx.msm(bm, g["scalars"], mh.FORM_MONT)
    assert (r.affine_std == g["expected"]).all()
    ctx.upload_bases(g["bases"], mh.FORM_STD)
    r2 = ctx.msm_resident(g["scalars"])
    assert (r2.affine_std == g["expected"]).all()
    # second scalar vector against the same resident bases
    s2 = orc.gen_scalars(77, 1024)
    r3 = ctx.msm_resident(s2)
    exp, einf, _ = orc.msm_pippenger(g["bases"], s2, orc.FORM_STD)
    assert (r3.affine_std == exp).all() and not r3.is_infinity
    # fewer scalars than resident bases -> truncated to the shorter (metal_msm.rs:652-656)
    r4 = ctx.msm_resident(s2[:100])
    exp, _, _ = orc.msm_pippenger(g["bases"][:100], s2[:100], orc.FORM_STD)
    assert (r4.affine_std == exp).all()


@pytest.mark.parametrize("wb,flags", [(0, 0), (0, mh.FLAG_NO_GLV), (13, 0), (16, mh.FLAG_NO_GLV), (8, mh.FLAG_UNSIGNED_DIGITS), (0, mh.FLAG_WINDOW_TABLE)])
def test_reference_zkey_points_in_mont_form(hk, wb, flags):
    """The G1 points the reference itself ships (its Groth16 proving key, tests/golden/zkey_g1_points.json: R = 2^256 Montgomery words, four
    entries at infinity) through the HIP path AS THEY ARE -- MSM_FORM_MONT, the word format of arkworks' Fq.0 -- against the pure-Python known
    answer; the same call on their standard-form images (converted by the DEVICE, fp op 4) gives the same words; so do the resident set, the
    arkworks-struct entry (72-byte records with the infinity byte) and plain sums with unit scalars.  VERDICT r4 item 3."""
    bases, inf, scalars, expected, d = load_zkey_points()
    with mh.MsmContext(window_bits=wb, flags=flags) as c:
        r = c.msm(bases, scalars, mh.FORM_MONT, inf)
        assert not r.is_infinity and (r.affine_std == expected).all()
        std = np.concatenate([hk.test_fp_op(4, bases[:, :8]), hk.test_fp_op(4, bases[:, 8:])], axis=1)  # Montgomery -> standard on the device
        assert (std[inf == 1] == 0).all()
        r2 = c.msm(std, scalars, mh.FORM_STD, inf)
        assert (r2.affine_std == expected).all()
        c.upload_bases(bases, mh.FORM_MONT, inf)
        assert (c.msm_resident(scalars).affine_std == expected).all()
        ones = np.zeros_like(scalars)
        ones[:, 0] = 1
        exp_sum = np.concatenate([np.frombuffer(int(v, 16).to_bytes(32, "little"), dtype="<u4") for v in d["expected_sum_of_points_affine_std_hex"]])
        assert (c.msm(bases, ones, mh.FORM_MONT, inf).affine_std == exp_sum).all()
        if not flags and not wb:
            # the struct entry: x | y | infinity byte, 72-byte stride, Fr in Montgomery form
            img = _ark_image(std, inf, 72, 0, 32, 64, np.random.default_rng(5))
            ra = c.msm_arkworks(img, 72, 0, 32, 64, _fr_mont(scalars))
            assert (ra.affine_std == expected).all()


@pytest.mark.parametrize("wb,flags", [(0, 0), (0, mh.FLAG_NO_GLV), (16, 0), (13, mh.FLAG_NO_GLV), (0, mh.FLAG_WINDOW_TABLE), (0, mh.FLAG_DETERMINISTIC)])
def test_reference_srs_known_answer_msms(wb, flags):
    """The MSM known answers the reference itself ships (tests/golden/srs_kzg_points.json: halo2 KZG parameters, monomial and Lagrange basis of one
    tau): g[j] = sum_i omega^(i*j) g_lagrange[i] for every j -- inputs AND expected outputs are reference-held R = 2^256 Montgomery words written by
    halo2curves.  Through the HIP path as they are (MSM_FORM_MONT), every j of both files, host call and resident set; the Jacobian result is
    compared PROJECTIVELY with the reference-held point as well (what the reference's own gate does, metal_msm.rs:739-760)."""
    for fname, k, omega, g, gl in load_srs_sets():
        n = 1 << k
        with mh.MsmContext(window_bits=wb, flags=flags) as c:
            c.upload_bases(gl, mh.FORM_MONT)
            for j in range(n):
                scalars = np.stack([orc.int_to_words(pow(omega, i * j, R)) for i in range(n)])
                exp = np.concatenate([orc.fq_from_mont(g[j, :8]), orc.fq_from_mont(g[j, 8:])])
                r = c.msm(gl, scalars, mh.FORM_MONT)
                assert not r.is_infinity and (r.affine_std == exp).all(), (fname, j)
                rr = c.msm_resident(scalars)
                if not (rr.affine_std == exp).all():  # (seen ONCE in ~20 runs of the suite in round 6: say whether the same call repeated gives the point)
                    again = [bool((c.msm_resident(scalars).affine_std == exp).all()) for _ in range(3)]
                    raise AssertionError((fname, j, "resident", "the same call repeated equals the expected point:", again, c.timings()))
                # X = x Z^2, Y = y Z^3 against the stored (x, y): projective equality with the reference-held point
                X, Y, Z = (orc.words_to_int(orc.fq_from_mont(r.jacobian_mont[8 * t:8 * t + 8])) for t in range(3))
                x, y = orc.words_to_int(exp[:8]), orc.words_to_int(exp[8:])
                assert Z != 0 and (X - x * Z * Z) % P == 0 and (Y - y * Z * Z * Z) % P == 0, (fname, j, "projective")


def test_reference_error_and_truncation_semantics(ctx):
    g = load_golden("rand_n17")
    with pytest.raises(mh.MsmError) as e:
        ctx.msm(np.zeros((0, 16), np.uint32), np.zeros((0, 8), np.uint32))
    assert str(e.value) == "Empty input" and e.value.code == mh.ERR_EMPTY
    with pytest.raises(mh.MsmError):
        mh.hip_variable_base_msm(g["bases"], np.zeros((0, 8), np.uint32))
    # unequal lengths -> min (metal_msm.rs:652-656)
    r = ctx.msm(g["bases"], g["scalars"][:9])
    exp, _, _ = orc.msm_naive(g["bases"][:9], g["scalars"][:9], orc.FORM_STD)
    assert (r.affine_std == exp).all()
    # non-canonical scalar (>= 2^254) is rejected, not silently mis-computed
    bad = g["scalars"].copy()
    bad[3, 7] |= 0x40000000
    with pytest.raises(mh.MsmError) as e:
        ctx.msm(g["bases"], bad)
    assert e.value.code == mh.ERR_BAD_ARG
    # the context stays usable after an error
    r = ctx.msm(g["bases"], g["scalars"])
    assert (r.affine_std == g["expected"]).all()
    assert mh.metal_variable_base_msm is mh.hip_variable_base_msm


def test_device_entry_with_infinity_mask_and_caller_stream(ctx, hk):
    """msm_bn254_g1_device(d_inf_mask != NULL, hip_stream != NULL) -- the two ABI arguments no other test passes.  Goldens with
    infinity masks on a caller-owned torch stream; then 2^19 points (above 2^18 the base conversion forks to the context's second
    stream behind an event recorded on the CALLER's stream) with an infinity mask, inputs still being produced on that stream
    when the call is made.  Mirrors the reference's e2e shape (tests/cuzk/e2e.rs:14-63): random instance, compare with the CPU."""
    import torch
    dev = torch.device("cuda:0")
    st = torch.cuda.Stream(device=dev)

    def to_dev(a):
        return torch.from_numpy(np.ascontiguousarray(a).view(np.int32).reshape(-1)).to(dev)

    for name in ("edge_inf_bases", "rand_n1024", "rand_n4096", "rand_n17"):
        g = load_golden(name)
        bm = np.concatenate([hk.test_fp_op(3, g["bases"][:, :8]), hk.test_fp_op(3, g["bases"][:, 8:])], axis=1)  # Montgomery words
        d_b, d_s = to_dev(bm), to_dev(g["scalars"])
        d_i = torch.from_numpy(np.ascontiguousarray(g["inf"], dtype=np.uint8)).to(dev)
        torch.cuda.synchronize()
        for stream in (None, st.cuda_stream):
            r = ctx.msm_device(d_b.data_ptr(), d_s.data_ptr(), g["bases"].shape[0], d_inf_ptr=d_i.data_ptr(), stream=stream)
            assert r.is_infinity == bool(g["expected_inf"]) and (r.affine_std == g["expected"]).all(), (name, stream)
    n = 1 << 19
    k = th.generate_scalars_host(0xB2540071, n, nonzero=True)
    s = th.generate_scalars_host(0xB2540072, n)
    inf = (np.arange(n) % 97 == 5).astype(np.uint8)
    keep = inf == 0
    exp, _ = orc.closed_form_expected(k[keep], s[keep])
    d_b = torch.empty(n * 16, dtype=torch.int32, device=dev)
    d_s = torch.empty(n * 8, dtype=torch.int32, device=dev)
    hk.generate_device(0xB2540071, 0xB2540072, n, d_b.data_ptr(), d_s.data_ptr())
    torch.cuda.synchronize()
    h_i = torch.from_numpy(inf).pin_memory()
    for rep in range(3):
        with torch.cuda.stream(st):
            # the mask and a copy of the scalars are PRODUCED on the caller's stream right before the call: the engine must order
            # its second stream behind them (event fork, msm_bn254_g1_device)
            d_i = h_i.to(dev, non_blocking=True)
            d_s2 = d_s.clone()
            d_b2 = d_b.clone()
        r = ctx.msm_device(d_b2.data_ptr(), d_s2.data_ptr(), n, d_inf_ptr=d_i.data_ptr(), stream=st.cuda_stream)
        assert (r.affine_std == exp).all() and not r.is_infinity, rep
    st.synchronize()
    # ... and msm_bn254_g1_multi_device with per-shard masks ({0,0}: two ranks on this GPU, host fold)
    with mh.MsmMulti(devices=[0, 0]) as m:
        h = n // 2
        r = m.msm_device([d_b.data_ptr(), d_b.data_ptr() + 64 * h], [d_s.data_ptr(), d_s.data_ptr() + 32 * h], [h, n - h],
                         d_inf_ptrs=[d_i.data_ptr(), d_i.data_ptr() + h])
        assert (r.affine_std == exp).all()
        r = m.msm_device([d_b.data_ptr(), d_b.data_ptr() + 64 * h], [d_s.data_ptr(), d_s.data_ptr() + 32 * h], [h, n - h],
                         d_inf_ptrs=[None, d_i.data_ptr() + h])  # nullable entries
        keep2 = keep.copy()
        keep2[:h] = True
        e2, _ = orc.closed_form_expected(k[keep2], s[keep2])
        assert (r.affine_std == e2).all()


@pytest.mark.parametrize("logn", [10, 12, 16])
def test_msm_random_vs_oracle(ctx, logn):
    n = 1 << logn
    k = orc.gen_scalars(0xB2540001, n, nonzero=True)
    s = orc.gen_scalars(0xB2540002 + logn, n)
    bases = orc.gen_bases_from_logs(k, orc.FORM_MONT)
    r = ctx.msm(bases, s, mh.FORM_MONT)
    exp, einf, _ = orc.msm_pippenger(bases, s, orc.FORM_MONT)
    assert not r.is_infinity and einf == 0 and (r.affine_std == exp).all()
    cf, _ = orc.closed_form_expected(k, s)
    assert (r.affine_std == cf).all()
    # determinism: same bits on a second launch
    r2 = ctx.msm(bases, s, mh.FORM_MONT)
    assert (r2.affine_std == r.affine_std).all()


def test_adversarial_distributions(ctx):
    """skewed digits (all-equal scalars), 1% duplicates, infinities, a (P,-P) pair -- SURVEY section 8d."""
    n = 1 << 12
    k = orc.gen_scalars(21, n, nonzero=True)
    k[100:140] = k[7]  # duplicates
    kneg = orc.int_to_words(R - orc.words_to_int(k[9]))
    k[10] = kneg
    bases = orc.gen_bases_from_logs(k, orc.FORM_MONT)
    inf = np.zeros(n, np.uint8)
    inf[[3, 500, 4095]] = 1
    for label, s in [("all-equal", np.tile(orc.gen_scalars(3, 1), (n, 1))), ("uniform", orc.gen_scalars(4, n)),
                     ("small", np.pad(orc.gen_scalars(5, n)[:, :1], ((0, 0), (0, 7)))), ("zeros", np.zeros((n, 8), np.uint32))]:
        s = s.copy()
        s[10] = s[9]
        r = ctx.msm(bases, s, mh.FORM_MONT, inf)
        exp, einf, _ = orc.msm_pippenger(bases, s, orc.FORM_MONT, inf)
        assert r.is_infinity == bool(einf) and (r.affine_std == exp).all(), label


def test_randomised_parity_fuzz():
    """40 random (size, window size, digit mode, infinity mask, scalar skew) combinations against the oracle
    (tools/fuzz_parity.py is the long-running version of the same net)."""
    rng = np.random.default_rng(20261002)
    nmax = 20000
    k_all = orc.gen_scalars(4242, nmax, nonzero=True)
    bases_all = orc.gen_bases_from_logs(k_all, orc.FORM_MONT)
    s_all = orc.gen_scalars(4343, nmax)
    for it in range(40):
        n = int(rng.choice([rng.integers(1, 40), rng.integers(40, 3000), rng.integers(3000, nmax)]))
        off = int(rng.integers(0, nmax - n + 1))
        bases = bases_all[off:off + n].copy()
        s = s_all[off:off + n].copy()
        mode = int(rng.integers(0, 8))
        if mode == 1:
            s[:] = s[0]
        elif mode == 2:
            s = s[np.arange(n) % 3]
        elif mode == 3:
            s[:, 1:] = 0
        elif mode == 4:
            s[rng.random(n) < 0.6] = 0
        elif mode == 5:
            u = rng.random(n)
            s[u < 0.7] = 0
            s[(u >= 0.3) & (u < 0.7), 0] = 1
        elif mode == 6:
            s[:] = orc.int_to_words(R - 1)
        elif mode == 7 and n > 1:
            bases[1::2] = bases[0]
        inf = (rng.random(n) < rng.choice([0.001, 0.05, 0.9])).astype(np.uint8) if rng.random() < 0.4 else None
        wb = int(rng.choice([0, 0, 0, 2, 3, 5, 8, 11, 12, 13, 14, 15, 16, 17, 18]))
        flags = mh.FLAG_UNSIGNED_DIGITS if (rng.random() < 0.25 and wb not in (17, 18)) else 0
        if rng.random() < 0.35:
            flags |= mh.FLAG_NO_GLV  # the unsplit (reference-shaped) pipeline
        with mh.MsmContext(window_bits=wb, flags=flags) as c:
            r = c.msm(bases, s, mh.FORM_MONT, inf)
        exp, einf, _ = orc.msm_pippenger(bases, s, orc.FORM_MONT, inf)
        assert r.is_infinity == bool(einf) and (r.affine_std == exp).all(), dict(case=it, n=n, mode=mode, wb=wb, flags=flags)


# ---- BASELINE.json full sizes: size-independent properties ---------------------------------------
@pytest.mark.parametrize("logn", [16, 20])
def test_full_size_closed_form_and_linearity(ctx, hk, logn):
    import torch
    n = 1 << logn
    dev = torch.device("cuda:0")
    d_bases = torch.empty(n * 16, dtype=torch.int32, device=dev)
    d_s = torch.empty(n * 8, dtype=torch.int32, device=dev)
    hk.generate_device(0xB2540001, 0xB2540002, n, d_bases.data_ptr(), d_s.data_ptr())
    torch.cuda.synchronize()
    # generated bases are k_i*G: spot-check against the oracle
    k = th.generate_scalars_host(0xB2540001, n, nonzero=True)
    s = th.generate_scalars_host(0xB2540002, n)
    hb = d_bases.cpu().numpy().view(np.uint32).reshape(n, 16)
    idx = [0, 1, 2, n // 2, n - 1, 12345 % n]
    assert (hb[idx] == orc.gen_bases_from_logs(k[idx], orc.FORM_MONT)).all()
    assert (d_s.cpu().numpy().view(np.uint32).reshape(n, 8) == s).all()
    r = ctx.msm_device(d_bases.data_ptr(), d_s.data_ptr(), n)
    exp, einf = orc.closed_form_expected(k, s)
    assert not r.is_infinity and (r.affine_std == exp).all()
    # linearity: MSM(B, s) + MSM(B, t) == MSM(B, s + t mod r), through the same HIP path
    t = th.generate_scalars_host(0xB2540003, n)
    d_t = torch.from_numpy(t.view(np.int32).reshape(-1)).to(dev)
    rt = ctx.msm_device(d_bases.data_ptr(), d_t.data_ptr(), n)
    to_int = lambda a: [sum(int(w) << (32 * j) for j, w in enumerate(row)) for row in a.tolist()]
    st = np.array([orc.int_to_words((x + y) % R) for x, y in zip(to_int(s), to_int(t))], dtype=np.uint32)
    d_st = torch.from_numpy(st.view(np.int32).reshape(-1)).to(dev)
    rst = ctx.msm_device(d_bases.data_ptr(), d_st.data_ptr(), n)
    comb = mh.combine_partials(np.stack([r.jacobian_mont, rt.jacobian_mont]))
    assert (comb.affine_std == rst.affine_std).all()
    # point-range sharding (the multi-GPU decomposition) gives the same group element
    h = n // 2
    p0 = ctx.msm_device(d_bases.data_ptr(), d_s.data_ptr(), h)
    assert ctx.timings()["accumulate_ms"] == 0  # no launch carries events unless asked for (msm_set_kernel_timing, ABI 7)
    ctx.set_kernel_timing(1)
    p1 = ctx.msm_device(d_bases.data_ptr() + h * 64, d_s.data_ptr() + h * 32, n - h)
    ctx.set_kernel_timing(0)
    comb = mh.combine_partials(np.stack([p0.jacobian_mont, p1.jacobian_mont]))
    assert (comb.affine_std == exp).all()
    tm = ctx.timings()
    assert tm["num_points"] == n - h and tm["accumulate_ms"] > 0 and tm["sort_ms"] == 0
    # ABI 7: the size-checked getter writes at most the caller's sizeof (a consumer built against an older, shorter msm_timings_t is not overrun)
    import ctypes as C
    buf = (C.c_uint8 * 96)(*([0xAB] * 96))
    assert ctx._lib.msm_get_timings_sized(ctx._h, C.cast(buf, C.c_void_p), 40) == mh.OK  # 40 bytes: h2d_ms .. total_ms + num_points
    assert bytes(buf[40:]) == b"\xab" * 56 and C.cast(buf, C.POINTER(C.c_float))[4] == pytest.approx(tm["accumulate_ms"])
    assert C.cast(C.byref(buf, 32), C.POINTER(C.c_uint64))[0] == n - h
    assert ctx._lib.msm_get_timings_sized(ctx._h, C.cast(buf, C.c_void_p), 0) == mh.ERR_BAD_ARG
    ctx.set_stage_timing(True)
    ctx.msm_device(d_bases.data_ptr(), d_s.data_ptr(), n)
    tm = ctx.timings()
    ctx.set_stage_timing(False)
    assert tm["sort_ms"] > 0 and tm["reduce_ms"] > 0 and tm["decompose_ms"] > 0


@pytest.mark.parametrize("logn", [17, 20])
def test_skewed_scalars_full_size_closed_form(ctx, hk, logn):
    """Skewed digit distributions at full size: buckets of up to N entries (thousands of chunks -> the segmented
    k_combine_long with its last-arriver finish, the direct-placement branch of the fine sort, wave-aggregated LDS
    counters).  Expected value by the closed form (sum s_i k_i) * G; every case twice (per-bucket counters self-clean)."""
    import torch
    n = 1 << logn
    dev = torch.device("cuda:0")
    d_bases = torch.empty(n * 16, dtype=torch.int32, device=dev)
    d_s = torch.empty(n * 8, dtype=torch.int32, device=dev)
    hk.generate_device(0xB2540011, 0xB2540012, n, d_bases.data_ptr(), d_s.data_ptr())
    torch.cuda.synchronize()
    k = th.generate_scalars_host(0xB2540011, n, nonzero=True)
    s = th.generate_scalars_host(0xB2540012, n)
    idx = np.arange(n)
    cases = {"all-equal": np.tile(s[:1], (n, 1)), "2-distinct": s[idx % 2], "3-blocked": s[(idx * 3) // n], "256-distinct": s[idx % 256],
             "below-2^32": np.pad(s[:, :1], ((0, 0), (0, 7))), "ones-and-zeros": np.pad((idx % 3 != 0).astype(np.uint32)[:, None], ((0, 0), (0, 7)))}
    for label, arr in cases.items():
        arr = np.ascontiguousarray(arr, dtype=np.uint32)
        exp, einf = orc.closed_form_expected(k, arr)
        d_t = torch.from_numpy(arr.view(np.int32).reshape(-1).copy()).to(dev)
        for rep in range(2):
            r = ctx.msm_device(d_bases.data_ptr(), d_t.data_ptr(), n)
            assert r.is_infinity == bool(einf) and (r.affine_std == exp).all(), (label, rep)


@pytest.mark.parametrize("logn", [18, 20])
def test_glv_split_full_size_matches_unsplit_and_closed_form(monkeypatch, hk, logn):
    """The GLV split (default up to 2^19 points, forced here at 2^20 too) against the unsplit pipeline and the closed form."""
    import torch
    n = 1 << logn
    dev = torch.device("cuda:0")
    d_bases = torch.empty(n * 16, dtype=torch.int32, device=dev)
    d_s = torch.empty(n * 8, dtype=torch.int32, device=dev)
    k = th.generate_scalars_host(0xB2540021, n, nonzero=True)
    s = th.generate_scalars_host(0xB2540022, n)
    exp, einf = orc.closed_form_expected(k, s)
    monkeypatch.setenv("MSM_HIP_GLV_MAX_LOG2", "23")  # a knob of the HOOKS build (the product library reads no planner knob)
    assert th.hooks_plan(n).glv == 1 and th.hooks_plan(n, 0, mh.FLAG_NO_GLV).glv == 0
    assert mh.plan(n).glv == (1 if n <= (1 << 20) else 0), "the product's planner must not read MSM_HIP_GLV_MAX_LOG2"
    with th.HooksContext() as cg, th.HooksContext(flags=mh.FLAG_NO_GLV) as cp:
        hk.generate_device(0xB2540021, 0xB2540022, n, d_bases.data_ptr(), d_s.data_ptr())
        torch.cuda.synchronize()
        rg = cg.msm_device(d_bases.data_ptr(), d_s.data_ptr(), n)
        rp = cp.msm_device(d_bases.data_ptr(), d_s.data_ptr(), n)
        assert (rg.affine_std == exp).all() and (rp.affine_std == exp).all() and not rg.is_infinity
        # resident bases: the phi records live behind the first n; a call on fewer scalars runs unsplit on the same set
        hb = d_bases.cpu().numpy().view(np.uint32).reshape(n, 16)
        cg.upload_bases(hb, mh.FORM_MONT)
        assert (cg.msm_resident(s).affine_std == exp).all()
        m = n // 3
        e2, _ = orc.closed_form_expected(k[:m], s[:m])
        assert (cg.msm_resident(s[:m]).affine_std == e2).all()


@pytest.mark.parametrize("flags", [0, mh.FLAG_WINDOW_TABLE, mh.FLAG_NO_GLV | mh.FLAG_WINDOW_TABLE])
def test_resident_bases_with_scalars_in_hbm(hk, flags):
    """msm_bn254_g1_resident_device (ABI 4): the scalars of a prover whose witness lives on the GPU -- resident bases (and their window
    table), device scalars, the caller's stream.  Same bits as the host-scalar call and the closed form; fewer scalars than bases run on
    the first n; no resident set -> MSM_ERR_STATE; NULL / empty -> the reference's errors (metal_msm.rs:647-656)."""
    import torch
    n = (1 << 17) + 77
    dev = torch.device("cuda:0")
    d_bases = torch.empty(n * 16, dtype=torch.int32, device=dev)
    d_s = torch.empty(n * 8, dtype=torch.int32, device=dev)
    k = th.generate_scalars_host(0xB2540071, n, nonzero=True)
    s = th.generate_scalars_host(0xB2540072, n)
    exp, einf = orc.closed_form_expected(k, s)
    hk.generate_device(0xB2540071, 0xB2540072, n, d_bases.data_ptr(), d_s.data_ptr())
    torch.cuda.synchronize()
    hb = d_bases.cpu().numpy().view(np.uint32).reshape(n, 16)
    with mh.MsmContext(flags=flags) as c:
        with pytest.raises(mh.MsmError) as e:
            c.msm_resident_device(d_s.data_ptr(), n)
        assert e.value.code == mh.ERR_STATE
        c.upload_bases(hb, mh.FORM_MONT)
        r = c.msm_resident_device(d_s.data_ptr(), n)
        assert (r.affine_std == exp).all() and r.is_infinity == bool(einf)
        assert (c.msm_resident(s).affine_std == exp).all()  # (Jacobian words are not compared: the order inside a bucket is the sort's atomics')
        # the caller's stream produced the scalars: they are written on a side stream right before the call
        side = torch.cuda.Stream(device=dev)
        d_t = torch.empty_like(d_s)
        with torch.cuda.stream(side):
            d_t.copy_(d_s, non_blocking=True)
            r2 = c.msm_resident_device(d_t.data_ptr(), n, stream=side.cuda_stream)
        assert (r2.affine_std == exp).all()
        m = n // 3
        e3, _ = orc.closed_form_expected(k[:m], s[:m])
        assert (c.msm_resident_device(d_s.data_ptr(), m).affine_std == e3).all()
        with pytest.raises(mh.MsmError) as e:
            c.msm_resident_device(d_s.data_ptr(), 0)
        assert e.value.code == mh.ERR_EMPTY
        with pytest.raises(mh.MsmError) as e:
            c.msm_resident_device(None, n)
        assert e.value.code == mh.ERR_BAD_ARG


@pytest.mark.parametrize("chunk_len", ["1", "2", "7", "26", "35", "1024"])
def test_forced_piece_lengths(monkeypatch, hk, chunk_len):
    """k_accumulate_pieces' work items: whole buckets up to pmax entries, runs of psplit entries of longer ones, counting-sorted by length
    (k_piece_count / k_piece_scatter), split buckets folded by k_combine_pieces (2..7 pieces: one thread; 8+: LDS trees, 2048-piece
    segments).  MSM_HIP_PIECE_LEN forces pmax = psplit: 1 (every entry its own piece, every bucket of two entries split), 2, 7, odd values,
    and 1024 (the largest bin; the long bucket below is still split) -- plain + GLV + window table + batch, against the closed form."""
    import torch
    monkeypatch.setenv("MSM_HIP_PIECE_LEN", chunk_len)
    n = (1 << 16) + 4321
    dev = torch.device("cuda:0")
    d_bases = torch.empty(n * 16, dtype=torch.int32, device=dev)
    d_s = torch.empty(n * 8, dtype=torch.int32, device=dev)
    k = th.generate_scalars_host(0xB2540061, n, nonzero=True)
    s = th.generate_scalars_host(0xB2540062, n)
    s[: n // 5] = s[0]          # a long bucket in every window
    exp, einf = orc.closed_form_expected(k, s)
    hk.generate_device(0xB2540061, 0xB2540062, n, d_bases.data_ptr(), d_s.data_ptr())
    d_s.copy_(torch.from_numpy(s.view(np.int32).reshape(-1)))
    torch.cuda.synchronize()
    hb = d_bases.cpu().numpy().view(np.uint32).reshape(n, 16)
    for flags in (0, mh.FLAG_NO_GLV, TABLE, TABLE | mh.FLAG_NO_GLV):
        with th.HooksContext(flags=flags) as c:  # (MSM_HIP_PIECE_LEN is a knob of the hooks build)
            r = c.msm_device(d_bases.data_ptr(), d_s.data_ptr(), n)
            assert (r.affine_std == exp).all() and r.is_infinity == bool(einf), (chunk_len, flags)
            c.upload_bases(hb, mh.FORM_MONT)
            for r in c.msm_resident_batch([s, s, s], want_affine=True):
                assert (r.affine_std == exp).all(), (chunk_len, flags, "batch")


# ---- row f4: the window table of a resident base set (MSM_FLAG_WINDOW_TABLE) --------------------------------------------------------
TABLE = mh.FLAG_WINDOW_TABLE


@pytest.mark.parametrize("wb,flags", [(0, TABLE), (0, TABLE | mh.FLAG_NO_GLV), (8, TABLE | mh.FLAG_NO_GLV), (13, TABLE), (16, TABLE),
                                      (20, TABLE | mh.FLAG_NO_GLV), (5, TABLE | mh.FLAG_NO_GLV)])
def test_window_table_golden_cases(wb, flags):
    """every golden case through upload_bases (+ table build) and msm_resident: one bucket array shared by all windows, digits of
    window j add +-T_j[i] = +-2^(c*j) P_i.  Planner plans (GLV split, c = 10 / 16) and forced widths incl. c = 20 (2^19 buckets, super-
    tile sort, pseudo-window reduction) and c = 5 (51 windows: long buckets in the shared array).  Edge cases of the goldens: bases at
    infinity, P and -P, equal bases, zero scalars, r - 1."""
    with mh.MsmContext(window_bits=wb, flags=flags) as c:
        for name in golden_cases():
            g = load_golden(name)
            n = g["bases"].shape[0]
            pl = mh.plan(n, wb, flags)
            assert pl.table_factor == pl.num_windows >= 2 and pl.bucket_arrays == 1 and pl.table_bytes == pl.table_factor * pl.virtual_points * 64
            c.upload_bases(g["bases"], mh.FORM_STD, g["inf"])
            r = c.msm_resident(g["scalars"])
            assert r.is_infinity == bool(g["expected_inf"]) and (r.affine_std == g["expected"]).all(), (name, wb, flags)
            if n > 2:  # fewer scalars than bases: the plain pipeline on the first records of level 0
                m = n - 1
                exp, einf, _ = orc.msm_pippenger(g["bases"][:m], g["scalars"][:m], orc.FORM_STD, g["inf"][:m])
                r = c.msm_resident(g["scalars"][:m])
                assert r.is_infinity == bool(einf) and (r.affine_std == exp).all(), (name, "truncated")
            # and the host-pointer entry of the same context is unaffected by the flag
            r = c.msm(g["bases"], g["scalars"], mh.FORM_STD, g["inf"])
            assert r.is_infinity == bool(g["expected_inf"]) and (r.affine_std == g["expected"]).all(), (name, "host call")


def test_window_table_collisions_across_windows():
    """what a shared bucket array adds to the complete group law's job: table records of DIFFERENT windows can be equal or opposite
    points.  Bases k*G and (2^c k)*G: T_1 of the first is T_0 of the second; with equal (opposite) digits in the two windows they meet
    in one bucket as P + P (P + (-P)).  Against the closed form."""
    c_bits, n = 8, 512
    k = orc.gen_scalars(0xB2540401, n, nonzero=True)
    ki = [orc.words_to_int(w) for w in k]
    for i in range(0, n, 2):  # odd points: 2^c times the point before
        ki[i + 1] = (ki[i] << c_bits) % R
    k = np.array([orc.int_to_words(v) for v in ki], dtype=np.uint32)
    bases = orc.gen_bases_from_logs(k, orc.FORM_MONT)
    s = orc.gen_scalars(0xB2540402, n)
    si = [orc.words_to_int(w) for w in s]
    for i in range(0, n, 2):
        d = (si[i] >> c_bits) & 0x7F  # window 1 of the even point (small enough to stay a positive digit; no carry from window 0) ...
        si[i] = (si[i] & ~(0xFF << c_bits) & ~0x80) | (d << c_bits)
        si[i + 1] = (si[i + 1] & ~0xFF) | ((d if i % 4 == 0 else (256 - d) & 0xFF))  # ... equals / negates window 0 of the odd one
    s = np.array([orc.int_to_words(v % R) for v in si], dtype=np.uint32)
    exp, einf = orc.closed_form_expected(k, s)
    with mh.MsmContext(window_bits=c_bits, flags=TABLE | mh.FLAG_NO_GLV) as c:
        c.upload_bases(bases, mh.FORM_MONT)
        r = c.msm_resident(s)
    assert r.is_infinity == bool(einf) and (r.affine_std == exp).all()


@pytest.mark.parametrize("logn,extra", [(16, 0), (19, 12345), (20, 0), (21, 0)])
def test_window_table_full_size_closed_form_batch_and_skew(hk, logn, extra):
    """the planner's table plans at full size against the closed form: 2^16 (GLV, c = 16, 8 windows in one array), 2^19 + 12345
    (unsplit c = 20: 13 windows x n entries cross the sort's super-tiles of 2^22 positions; infinity mask), 2^20 (the BASELINE
    size) and 2^21 (the largest size that gets a table: 27 M entries in one sort, every region of the fine sort 1.6 staging areas long and
    placed directly by its owner, with the super-tile recovery).  Single resident calls, the batch entry (two MSMs in flight), a truncated call, all-equal scalars (13 buckets hold everything:
    oversized regions with 512 fine bins), and the same set without the table as the reference bits."""
    import torch
    n = (1 << logn) + extra
    seed = 0xB2540410 + logn
    d_b = torch.empty(n * 16, dtype=torch.int32, device="cuda:0")
    d_s = torch.empty(n * 8, dtype=torch.int32, device="cuda:0")
    hk.generate_device(seed, seed + 1, n, d_b.data_ptr(), d_s.data_ptr())
    torch.cuda.synchronize()
    hb = d_b.cpu().numpy().view(np.uint32).reshape(n, 16)
    k = th.generate_scalars_host(seed, n, nonzero=True)
    vecs = [th.generate_scalars_host(seed + 1 + j, n) for j in range(3)]
    if logn == 20:
        # regression (round 3): the staged sort element of (window 3, point n - 1) is position 2^22 - 1 of its super-tile; give it the digit
        # -512 (bucket 511: all nine fine bits set, negative) and the element is the word 0xFFFFFFFF -- which the fine sort used to read
        # as "no element" and drop
        si = orc.words_to_int(vecs[2][n - 1])
        si = (si & ~(((1 << 40) - 1) << 40)) | (((1 << 20) - 512) << 60)
        vecs[2][n - 1] = orc.int_to_words(si)
    inf = (np.arange(n) % 1013 == 7).astype(np.uint8) if extra else None
    keep = slice(None) if inf is None else inf == 0
    exp = [orc.closed_form_expected(k[keep], v[keep])[0] for v in vecs]
    pl = mh.plan(n, 0, TABLE)
    assert pl.table_factor == pl.num_windows and pl.bucket_arrays == 1
    assert (pl.window_bits, pl.glv) == ((16, 1) if logn <= 18 else (20, 0))
    with mh.MsmContext(flags=TABLE) as c, mh.MsmContext() as plain:
        c.upload_bases(hb, mh.FORM_MONT, inf)
        plain.upload_bases(hb, mh.FORM_MONT, inf)
        for v, e in zip(vecs, exp):
            r = c.msm_resident(v)
            assert not r.is_infinity and (r.affine_std == e).all()
        res = c.msm_resident_batch(vecs + vecs)
        for j, r in enumerate(res):
            assert (r.affine_std == exp[j % 3]).all(), ("batch", j)
        m = n - 4321
        assert (c.msm_resident(vecs[0][:m]).affine_std == plain.msm_resident(vecs[0][:m]).affine_std).all()
        eq = np.tile(vecs[1][5], (n, 1))  # every scalar equal: W buckets of the one array hold all n*W entries
        assert (c.msm_resident(eq).affine_std == plain.msm_resident(eq).affine_std).all()
        small = vecs[2].copy()
        small[:, 1:] = 0                   # 32-bit scalars: only the two lowest windows are populated
        assert (c.msm_resident(small).affine_std == plain.msm_resident(small).affine_std).all()
        bad = vecs[0].copy()
        bad[n // 3, 7] = 0x40000000
        with pytest.raises(mh.MsmError) as e:
            c.msm_resident(bad)
        assert e.value.code == mh.ERR_BAD_ARG
        assert (c.msm_resident(vecs[0]).affine_std == exp[0]).all()  # the context and its table survive an error


def test_window_table_compressed_upload_and_memory_cap(monkeypatch):
    """msm_bn254_g1_upload_compressed builds the table behind the decoded points; MSM_HIP_TABLE_MAX_GB=0 (hooks build, read when the context
    is created) leaves a set without table and the calls on the ordinary path"""
    g = load_golden("rand_n4096")
    images = mh.compress_points(g["bases"], mh.FORM_STD, g["inf"])
    with mh.MsmContext(flags=TABLE) as c:
        c.upload_compressed(images)
        r = c.msm_resident(g["scalars"])
        assert (r.affine_std == g["expected"]).all()
    monkeypatch.setenv("MSM_HIP_TABLE_MAX_GB", "0")  # (hooks build; the product's cap is the built-in 64 GB)
    assert th.hooks_plan(4096, 0, TABLE).table_factor == 1 and mh.plan(4096, 0, TABLE).table_factor > 1
    with th.HooksContext(flags=TABLE) as c:
        c.upload_bases(g["bases"], mh.FORM_STD, g["inf"])
        assert (c.msm_resident(g["scalars"]).affine_std == g["expected"]).all()


# ---- BASELINE config 5: streamed host->HBM chunks overlapped with the pipeline ---------------------
def test_streamed_chunks_match_oracle(hk):
    """msm_bn254_g1 cuts n >= 2*chunk points into chunks (double-buffered H2D on a copy stream; every chunk accumulates INTO the
    shared bucket array, k_accumulate<true, true>: one bucket reduction and one host finish per MSM).  Forced to tiny chunks here so
    that the golden cases exercise it; ragged last chunk and infinity masks included."""
    with mh.MsmContext(stream_chunk_log2=8) as c:
        for name in ("rand_n1024", "rand_n4096"):
            g = load_golden(name)
            r = c.msm(g["bases"], g["scalars"], mh.FORM_STD, g["inf"])
            assert (r.affine_std == g["expected"]).all() and not r.is_infinity, name
            assert c.timings()["num_points"] == g["bases"].shape[0]
        g = load_golden("rand_n4096")
        n = 3000  # 11 full chunks + a ragged one
        inf = np.zeros(n, np.uint8)
        inf[[0, 255, 256, 2999]] = 1
        r = c.msm(g["bases"][:n], g["scalars"][:n], mh.FORM_STD, inf)
        exp, einf, _ = orc.msm_pippenger(g["bases"][:n], g["scalars"][:n], orc.FORM_STD, inf)
        assert (r.affine_std == exp).all() and r.is_infinity == bool(einf)
        # Montgomery-form input through the streamed path
        bm = np.concatenate([hk.test_fp_op(3, g["bases"][:, :8]), hk.test_fp_op(3, g["bases"][:, 8:])], axis=1)
        r = c.msm(bm, g["scalars"], mh.FORM_MONT)
        assert (r.affine_std == g["expected"]).all()
        # a non-canonical scalar in a LATER chunk is still rejected
        bad = g["scalars"].copy()
        bad[4000, 7] |= 0x40000000
        with pytest.raises(mh.MsmError) as e:
            c.msm(g["bases"], bad)
        assert e.value.code == mh.ERR_BAD_ARG
        # below the threshold the single-shot path is taken
        g = load_golden("rand_n256")
        r = c.msm(g["bases"], g["scalars"])
        assert (r.affine_std == g["expected"]).all()


def test_streamed_equals_single_shot_at_2_pow_18(ctx, hk):
    n = 1 << 18
    k = th.generate_scalars_host(0xB2540001, n, nonzero=True)
    s = th.generate_scalars_host(0xB2540009, n)
    import torch
    d_b = torch.empty(n * 16, dtype=torch.int32, device="cuda:0")
    d_s = torch.empty(n * 8, dtype=torch.int32, device="cuda:0")
    hk.generate_device(0xB2540001, 0xB2540009, n, d_b.data_ptr(), d_s.data_ptr())
    hb = d_b.cpu().numpy().view(np.uint32).reshape(n, 16)
    single = ctx.msm(hb, s, mh.FORM_MONT)
    with mh.MsmContext(stream_chunk_log2=15) as c:  # 8 chunks of 32768
        streamed = c.msm(hb, s, mh.FORM_MONT)
    exp, _ = orc.closed_form_expected(k, s)
    assert (single.affine_std == exp).all() and (streamed.affine_std == exp).all()


# ---- zero-copy arkworks ingestion -------------------------------------------
def _ark_image(bases_std, inf, stride, x_off, y_off, inf_off, rng):
    """byte image of a [G1Affine] slice: Fq Montgomery limbs at x_off / y_off, `infinity` bool at inf_off, garbage in
    the padding (arkworks' identity is x = y = 0, infinity = true)"""
    n = bases_std.shape[0]
    img = rng.integers(0, 256, size=(n, stride), dtype=np.uint8)
    R = 1 << 256
    for i in range(n):
        x = orc.words_to_int(bases_std[i, :8]) * R % P
        y = orc.words_to_int(bases_std[i, 8:]) * R % P
        if inf is not None and inf[i]:
            x = y = 0
        img[i, x_off:x_off + 32] = np.frombuffer(x.to_bytes(32, "little"), np.uint8)
        img[i, y_off:y_off + 32] = np.frombuffer(y.to_bytes(32, "little"), np.uint8)
        if inf_off is not None:
            img[i, inf_off] = 1 if (inf is not None and inf[i]) else 0
    return img


def _fr_mont(scalars):
    R = 1 << 256
    return np.stack([orc.int_to_words(orc.words_to_int(s) * R % orc.R_ORDER) for s in scalars])


@pytest.mark.parametrize("layout", [(72, 0, 32, 64), (80, 40, 8, 72), (64, 32, 0, None)])
def test_arkworks_struct_ingestion(ctx, layout):
    stride, x_off, y_off, inf_off = layout
    rng = np.random.default_rng(5)
    for name in ("rand_n17", "rand_n1024", "edge_inf_bases", "edge_scalar_r_minus_1", "edge_zero_scalars", "edge_p_minus_p"):
        g = load_golden(name)
        inf = g["inf"] if inf_off is not None else None
        if inf_off is None and g["inf"].any():
            continue
        img = _ark_image(g["bases"], inf, stride, x_off, y_off, inf_off, rng)
        r = ctx.msm_arkworks(img, stride, x_off, y_off, inf_off, _fr_mont(g["scalars"]))
        assert r.is_infinity == bool(g["expected_inf"]), (name, layout)
        assert (r.affine_std == g["expected"]).all(), (name, layout)


@pytest.mark.parametrize("slow", [False, True])
def test_arkworks_structs_optimistic_path_and_its_fallback(monkeypatch, slow):
    """round 6: a struct array is first taken as free of points at infinity (k_ark_repack: the words are repacked and gathered as they are, the sort starts with
    the scalars); a set `infinity` flag raises error bit 16 and the call is repeated through k_import_ark with the flags as a mask.  Same bits either way, with
    and without flags, single shot and streamed (2^19 points), and forced through the slow path from the start (MSM_HIP_ARK_SLOW of the hooks build)."""
    if slow:
        monkeypatch.setenv("MSM_HIP_ARK_SLOW", "1")
    n = 1 << 19
    k = th.generate_scalars_host(0xB2540D01, n, nonzero=True)
    s = th.generate_scalars_host(0xB2540D02, n)
    bases = orc.gen_bases_from_logs(k, orc.FORM_MONT)
    rinv = pow(1 << 256, -1, orc.R_ORDER)
    g = np.zeros(16, np.uint32)
    g[0], g[8] = 1, 2
    expect = lambda d: orc.g1_to_affine_std(orc.g1_scalar_mul(g, orc.int_to_words(d % orc.R_ORDER)))[0]
    img = np.zeros((n, 72), np.uint8)
    img[:, :64] = bases.view(np.uint8).reshape(n, 64)
    e_all = expect(orc.dot_words(k, s) * rinv)
    inf = np.zeros(n, np.uint8)
    inf[[0, 3, 77777, (1 << 18) - 1, 1 << 18, n - 1]] = 1
    img_inf = img.copy()
    img_inf[:, 64] = inf
    img_inf[inf != 0, :64] = 0  # arkworks' identity: x = y = 0, infinity = true
    s_masked = s.copy()
    s_masked[inf != 0] = 0
    e_inf = expect(orc.dot_words(k, s_masked) * rinv)
    with th.HooksContext() as c:
        for m in (n, 1 << 16, 4097):  # streamed (>= 2^19 points), single shot with the bases on the copy stream, and just above the overlap threshold
            e1 = e_all if m == n else expect(orc.dot_words(k[:m], s[:m]) * rinv)
            assert (c.msm_arkworks(img[:m], 72, 0, 32, 64, s[:m]).affine_std == e1).all(), (m, slow)
        assert c.timings()["stream_chunks"] == 0
        r = c.msm_arkworks(img_inf, 72, 0, 32, 64, s)  # flags set: the optimistic pass notices, the call is repeated
        assert (r.affine_std == e_inf).all() and c.timings()["stream_chunks"] >= 2
        assert (c.msm_arkworks(img, 72, 0, 32, 64, s).affine_std == e_all).all()  # and the context is as good as new
    with th.HooksContext(flags=mh.FLAG_NO_GLV, window_bits=13) as c:  # an unsplit plan: no phi records behind the repacked words
        assert (c.msm_arkworks(img_inf, 72, 0, 32, 64, s).affine_std == e_inf).all()
        assert (c.msm_arkworks(img, 72, 0, 32, 64, s).affine_std == e_all).all()


def test_arkworks_ingestion_rejects_bad_layout(ctx):
    g = load_golden("rand_n3")
    img = _ark_image(g["bases"], None, 72, 0, 32, 64, np.random.default_rng(1))
    for bad in [(72, 2, 32, 64), (72, 0, 48, 64), (60, 0, 28, 56), (72, 0, 32, 72)]:
        with pytest.raises(mh.MsmError) as e:
            ctx.msm_arkworks(img, bad[0], bad[1], bad[2], bad[3], _fr_mont(g["scalars"]))
        assert e.value.code == mh.ERR_BAD_ARG


@pytest.mark.parametrize("wb,flags", [(2, 0), (3, 0), (4, mh.FLAG_UNSIGNED_DIGITS), (17, 0), (18, 0), (20, 0)])
def test_extreme_window_sizes(wb, flags):
    """c = 2 (128 windows of 2 buckets) up to c = 20 (13 windows of 2^19 buckets, global-atomic sort fallback)."""
    with mh.MsmContext(window_bits=wb, flags=flags) as c:
        for name in ("rand_n1", "rand_n17", "rand_n256", "edge_inf_bases", "edge_p_minus_p", "edge_carry_patterns", "edge_scalar_r_minus_1"):
            g = load_golden(name)
            r = c.msm(g["bases"], g["scalars"], mh.FORM_STD, g["inf"])
            assert r.is_infinity == bool(g["expected_inf"]), (name, wb)
            assert (r.affine_std == g["expected"]).all(), (name, wb)


def test_repeatability_and_context_isolation():
    """the sort is not order-stable across launches (LDS atomics), the group element must be; two contexts at once"""
    g = load_golden("rand_n4096")
    with mh.MsmContext() as c1, mh.MsmContext(window_bits=11) as c2:
        outs = set()
        for _ in range(5):
            outs.add(c1.msm(g["bases"], g["scalars"]).affine_std.tobytes())
            outs.add(c2.msm(g["bases"], g["scalars"]).affine_std.tobytes())
        assert outs == {g["expected"].tobytes()}


def test_deterministic_flag_gives_one_jacobian_representation(hk):
    """include/msm_hip.h "Determinism": out_affine_std is canonical everywhere; out_jacobian_mont is a projective representative that may differ
    between identical calls (the sort places the entries of a bucket with LDS atomics) UNLESS the context carries MSM_FLAG_DETERMINISTIC, which
    hands out the Z = 1 representative: 20 identical calls => one set of 24 words, on every entry point, and Z is Montgomery one.
    Reference: metal_msm.rs:228-241 builds G::new(x, y, z) from a SERIAL sort (transpose.metal:8-65), so its limbs repeat."""
    import torch
    one = orc.fq_to_mont(orc.int_to_words(1))
    n = 1 << 14
    k = orc.gen_scalars(0xD37, n, nonzero=True)
    s = orc.gen_scalars(0xD38, n)
    bases = orc.gen_bases_from_logs(k, orc.FORM_MONT)
    exp, einf, _ = orc.msm_pippenger(bases, s, orc.FORM_MONT)
    for flags in (mh.FLAG_DETERMINISTIC, mh.FLAG_DETERMINISTIC | mh.FLAG_NO_GLV, mh.FLAG_DETERMINISTIC | mh.FLAG_WINDOW_TABLE):
        with mh.MsmContext(flags=flags) as c:
            c.upload_bases(bases, mh.FORM_MONT)
            d_b = torch.from_numpy(bases.view(np.int32)).cuda()
            d_s = torch.from_numpy(s.view(np.int32)).cuda()
            torch.cuda.synchronize()
            reps = set()
            for it in range(20):
                r = (c.msm(bases, s, mh.FORM_MONT), c.msm_resident(s), c.msm_device(d_b.data_ptr(), d_s.data_ptr(), n))[it % 3]
                assert (r.affine_std == exp).all() and not r.is_infinity
                assert (r.jacobian_mont[16:] == one).all(), "Z is not Montgomery one"
                aff, inf = orc.g1_to_affine_std(r.jacobian_mont)
                assert inf == 0 and (aff == exp).all()
                reps.add(r.jacobian_mont.tobytes())
            assert len(reps) == 1, (flags, len(reps))
            # the identity: (R, R, 0)
            z = c.msm(bases[:8], np.zeros((8, 8), np.uint32), mh.FORM_MONT)
            assert z.is_infinity and (z.jacobian_mont[:8] == one).all() and (z.jacobian_mont[8:16] == one).all() and not z.jacobian_mont[16:].any()
    # without the flag: the affine words are canonical; the Jacobian words are only required to BE the same group element
    with mh.MsmContext() as c:
        for _ in range(5):
            r = c.msm(bases, s, mh.FORM_MONT)
            aff, inf = orc.g1_to_affine_std(r.jacobian_mont)
            assert (r.affine_std == exp).all() and inf == 0 and (aff == exp).all()


def test_a_call_that_fails_between_sort_and_accumulation_leaves_the_context_usable():
    """the piece-sort histogram and its bin cursors were cleaned by workgroup 0 of the accumulation of the call that used
    them; a recoverable error between enqueue_sort and enqueue_accumulate (a failed copy or event wait in the overlap branch of the host call)
    left them dirty, and the next call's piece list was built from wrong bin starts -- silent corruption.  They are now zeroed at the head of
    every chain (k_decompose* / k_coarse_hist, msm_kernels.hpp clear_piece_bins).  The hook abandons a call exactly there -- on skewed scalars,
    whose long buckets fill many bins and both lists -- at several sizes; every following call on the same context must be right, including
    one of a different shape."""
    g = load_golden("rand_n4096")
    n = g["scalars"].shape[0]
    skew = g["scalars"].copy()
    skew[: n // 2] = skew[0]
    with th.HooksContext() as c:
        for cut in (n, n // 3, 257):
            for sc in (skew[:cut], g["scalars"][:cut]):
                with pytest.raises(mh.MsmError) as e:
                    c.abandon_after_sort(sc)
                assert e.value.code == mh.ERR_HIP
                r = c.msm(g["bases"], g["scalars"], mh.FORM_STD, g["inf"])
                assert r.is_infinity == bool(g["expected_inf"]) and (r.affine_std == g["expected"]).all(), cut
                small = load_golden("rand_n256")
                assert (c.msm(small["bases"], small["scalars"], mh.FORM_STD, small["inf"]).affine_std == small["expected"]).all()
