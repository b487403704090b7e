"""GPU tests (-m gpu): the reference's OWN fixture shape at full size.

`test_utils::generate_random_bases_and_scalars(size)` (metal_msm.rs:706-730) seeds every rayon thread with the same `test_rng()`:
the instance it returns is ONE (base, scalar) sequence of length size / T repeated T times, T = rayon's thread count -- every base AND
every scalar occurs T times, so every bucket that holds a point holds T copies of it and the bucket accumulation runs P + P (the
doubling branch of the mixed addition, ec_bn254.hpp xyzz_madd / xyzz_madd_m32) at every size the reference tests (2^16 in
metal_msm.rs:739-760, 2^16 ... 2^24 in benches/e2e.rs).  Here: the same shape built from the synthetic generator -- a sequence
(k_i * G, s_i), i < n / T, tiled T = 8 (a laptop) and T = 128 (the GPU box's host) times -- through the device call, the host-pointer call
and the resident set at 2^16 and 2^20, against the closed form (T * sum s_i k_i mod r) * G.

Second shape: the points the reference tree itself holds (15 zkey + 48 KZG-parameter points, tests/golden/*.json) tiled to 2^16 with
random scalars; expected = sum_j (sum_{i = j mod 63} s_i) * P_j from 63 oracle scalar multiplications."""
import numpy as np
import pytest

import mopro_msm_hip as mh
from mopro_msm_hip import testhooks as th
from conftest import load_srs_sets, load_zkey_points
from oracle import bn254_oracle as orc

pytestmark = pytest.mark.gpu
R = orc.R_ORDER


def _expected(dot):
    g = np.zeros(16, np.uint32)
    g[0], g[8] = 1, 2
    return orc.g1_to_affine_std(orc.g1_scalar_mul(g, orc.int_to_words(dot % R)))


@pytest.fixture(scope="module")
def hk():
    c = th.HooksContext()
    yield c
    c.close()


def tiled_instance(hk, logn, T, seed):
    """(d_bases, d_scalars, closed-form dot) of the reference's fixture shape: a sequence of n / T pairs repeated T times, in HBM"""
    import torch
    n = 1 << logn
    L = n // T
    sb = torch.empty(L * 16, dtype=torch.int32, device="cuda:0")
    ss = torch.empty(L * 8, dtype=torch.int32, device="cuda:0")
    hk.generate_device(seed, seed + 1, L, sb.data_ptr(), ss.data_ptr())
    torch.cuda.synchronize()
    d_b = sb.view(L, 16).repeat(T, 1).contiguous().view(-1)
    d_s = ss.view(L, 8).repeat(T, 1).contiguous().view(-1)
    k = th.generate_scalars_host(seed, L, nonzero=True)
    s = th.generate_scalars_host(seed + 1, L)
    return d_b, d_s, T * orc.dot_words(k, s)


@pytest.mark.parametrize("logn,T", [(16, 8), (16, 128), (17, 128), (18, 32), (19, 128), (20, 8), (20, 128)])  # (17 .. 19: SPARSE instances -- the kernels raise pmax, msmplan::effective_pmax)
def test_reference_fixture_shape_every_pair_repeated_T_times(hk, logn, T):
    n = 1 << logn
    d_b, d_s, dot = tiled_instance(hk, logn, T, 0xB2540F01 + 16 * logn + T)
    exp, einf = _expected(dot)
    assert einf == 0
    hb_t, hs_t = d_b.cpu(), d_s.cpu()
    hb = hb_t.numpy().view(np.uint32).reshape(n, 16)
    hs = hs_t.numpy().view(np.uint32).reshape(n, 8)
    assert (hb[: n // T] == hb[n - n // T:]).all() and (hs[: n // T] == hs[n // T: 2 * (n // T)]).all()  # the shape itself
    with mh.MsmContext() as c:
        for _ in range(2):  # device call (the bench line's call)
            r = c.msm_device(d_b.data_ptr(), d_s.data_ptr(), n)
            assert not r.is_infinity and (r.affine_std == exp).all()
        assert c.timings()["num_adds"] > 7 * n  # nothing was dropped as a duplicate
        r = c.msm(hb, hs, mh.FORM_MONT)  # host-pointer call (streamed from 2^19 points): the reference's own call shape
        assert (r.affine_std == exp).all()
        c.upload_bases(hb, mh.FORM_MONT)  # resident set (converted to the internal domain: the other mixed addition)
        assert (c.msm_resident(hs).affine_std == exp).all()
        assert (c.msm_resident_device(d_s.data_ptr(), n).affine_std == exp).all()
    if logn == 16:
        for kw in (dict(flags=mh.FLAG_NO_GLV), dict(flags=mh.FLAG_NO_GLV | mh.FLAG_UNSIGNED_DIGITS, window_bits=16), dict(window_bits=13),
                   dict(flags=mh.FLAG_WINDOW_TABLE)):
            with mh.MsmContext(**kw) as c:
                assert (c.msm(hb, hs, mh.FORM_MONT).affine_std == exp).all(), kw
                c.upload_bases(hb, mh.FORM_MONT)
                assert (c.msm_resident(hs).affine_std == exp).all(), kw


def test_identical_scalars_on_identical_bases_cancel_and_double(hk):
    """the shape pushed further: T = 2 with the second copy's scalars NEGATED (r - s_i) gives the identity (every bucket meets P and -P at
    every size), and with the second copy's scalars equal the double of the half-instance"""
    import torch
    logn, T = 18, 2
    n = 1 << logn
    d_b, d_s, dot = tiled_instance(hk, logn, T, 0xB2540F77)
    hs = d_s.cpu().numpy().view(np.uint32).reshape(n, 8).copy()
    half = n // 2
    neg = np.stack([orc.int_to_words((R - orc.words_to_int(hs[i])) % R) for i in range(0, half, 4099)])  # spot rows only (Python ints are slow) ...
    s2 = hs.copy()
    # ... the full negation by numpy: r - s as 8 x 32-bit words with borrow
    rw = np.array(orc.int_to_words(R), np.int64)
    a = hs[:half].astype(np.int64)
    out = np.zeros_like(a)
    borrow = np.zeros(half, np.int64)
    for j in range(8):
        d = rw[j] - a[:, j] - borrow
        borrow = (d < 0).astype(np.int64)
        out[:, j] = d + (borrow << 32)
    zero = (a == 0).all(axis=1)
    out[zero] = 0
    s2[half:] = out.astype(np.uint32)
    assert (s2[half:][::4099][: len(neg)] == neg).all()
    t = torch.from_numpy(s2.view(np.int32).reshape(-1).copy()).cuda()
    with mh.MsmContext() as c:
        r = c.msm_device(d_b.data_ptr(), t.data_ptr(), n)
        assert r.is_infinity and (r.affine_std == 0).all()
        r = c.msm_device(d_b.data_ptr(), d_s.data_ptr(), n)
        exp, _ = _expected(dot)
        assert (r.affine_std == exp).all()


def test_reference_held_points_tiled_to_2_pow_16():
    """the 15 curve points of the reference's zkey + the 48 points of its two KZG parameter files (R = 2^256 Montgomery words as stored),
    tiled to 2^16 bases with seeded random scalars: 63 distinct bases, each 1040 times -- buckets full of copies of the SAME third-party
    points.  Expected value from 63 oracle scalar multiplications of the column sums."""
    zb, zinf, _, _, _ = load_zkey_points()
    pts = [zb[i] for i in range(len(zb)) if not zinf[i]]
    for _f, _k, _om, g, gl in load_srs_sets():
        pts += list(g) + list(gl)
    pts = np.stack(pts).astype(np.uint32)
    assert pts.shape == (63, 16)
    n = 1 << 16
    idx = np.arange(n) % 63
    bases = np.ascontiguousarray(pts[idx])
    scalars = orc.gen_scalars(0xB2540F91, n)
    # column sums sum_{i = j mod 63} s_i mod r (numpy on 16-bit halves, exact), then 63 scalar multiplications and 62 additions on the oracle
    acc = None
    for j in range(63):
        col = scalars[idx == j]
        ones = np.zeros_like(col)
        ones[:, 0] = 1
        sj = orc.dot_words(col, ones) % R
        std = np.concatenate([orc.fq_from_mont(pts[j, :8]), orc.fq_from_mont(pts[j, 8:])])
        term = orc.g1_scalar_mul(std, orc.int_to_words(sj))
        acc = term if acc is None else orc.g1_add(acc, term)
    exp, einf = orc.g1_to_affine_std(acc)
    assert einf == 0
    for kw in (dict(), dict(flags=mh.FLAG_NO_GLV), dict(window_bits=13), dict(flags=mh.FLAG_WINDOW_TABLE)):
        with mh.MsmContext(**kw) as c:
            r = c.msm(bases, scalars, mh.FORM_MONT)
            assert not r.is_infinity and (r.affine_std == exp).all(), kw
            c.upload_bases(bases, mh.FORM_MONT)
            assert (c.msm_resident(scalars).affine_std == exp).all(), kw


@pytest.mark.parametrize("kind", ["witness-like", "below 2^32", "one hot value", "three distinct"])
def test_full_size_instances_that_sort_few_or_lumpy_entries(hk, kind):
    """Scalars that are mostly zero or tiny (a witness vector), or a handful of values: the instance sorts far fewer entries than its plan allows for and one
    or two buckets hold hundreds of thousands of them.  The kernels that cut such buckets choose the run length from the entries they really sorted
    (msmplan::effective_psplit; round 6, profiles/NOTES_r6.md section 14) while the workspace was sized on the host: 2^20 points through the device call,
    the streamed host call and the resident set against the closed form, and the list counters say that the long-bucket path was taken."""
    import torch
    n = 1 << 20
    d_b = torch.empty(n * 16, dtype=torch.int32, device="cuda:0")
    d_s = torch.empty(n * 8, dtype=torch.int32, device="cuda:0")
    hk.generate_device(0xB2540F71, 0xB2540F72, n, d_b.data_ptr(), d_s.data_ptr())
    torch.cuda.synchronize()
    k = th.generate_scalars_host(0xB2540F71, n, nonzero=True)
    s = d_s.cpu().numpy().view(np.uint32).reshape(n, 8).copy()
    rng = np.random.default_rng(7)
    u = rng.random(n)
    if kind == "witness-like":  # 40 % zeros, 30 % ones, 10 % below 2^16, 20 % uniform
        full = s.copy()
        s[u < 0.8] = 0
        s[(u >= 0.4) & (u < 0.7), 0] = 1
        sel = (u >= 0.7) & (u < 0.8)
        s[sel, 0] = full[sel, 0] & 0xFFFF
    elif kind == "below 2^32":
        s[:, 1:] = 0
    elif kind == "one hot value":  # 95 % of the scalars are one 254-bit value, the rest uniform
        s[u < 0.95] = s[0]
    else:
        s = s[np.arange(n) % 3].copy()
    exp, einf = _expected(orc.dot_words(k, s))
    assert einf == 0
    d_s2 = torch.from_numpy(s.view(np.int32).reshape(-1).copy()).cuda()
    hb = d_b.cpu().numpy().view(np.uint32).reshape(n, 16)
    for r in (hk.msm_device(d_b.data_ptr(), d_s2.data_ptr(), n), hk.msm_device(d_b.data_ptr(), d_s2.data_ptr(), n)):
        assert not r.is_infinity and (r.affine_std == exp).all(), kind
    counts = hk.list_counts()
    assert counts["long"] > 0, (kind, counts)  # buckets of eight or more pieces were folded
    assert (hk.msm(hb, s, mh.FORM_MONT).affine_std == exp).all(), kind  # streamed: every chunk picks its own run length
    with mh.MsmContext(flags=mh.FLAG_NO_GLV) as c:  # an unsplit plan: twice the windows, other bucket sizes
        assert (c.msm_device(d_b.data_ptr(), d_s2.data_ptr(), n).affine_std == exp).all(), kind
    hk.upload_bases(hb, mh.FORM_MONT)
    assert (hk.msm_resident(s).affine_std == exp).all(), kind
