"""Stage-level GPU parity (-m gpu): every stage of the pipeline against its own specification, through the stage-dump hook of
the hooks build (include/msm_hip_testhooks.h: msm_test_stage_dump runs the whole pipeline once and copies the intermediates of
each stage back).

Counterparts of the reference's per-kernel tests:
  decompose  tests/cuzk/convert_point_coords_and_decompose_scalars.rs:177-235   digits rebuild the scalar, |d| <= H
  sort       tests/cuzk/transpose.rs:6-118     offsets = exclusive prefix of the digit histogram; `sorted` is a permutation
                                               of the non-zero digits grouped by bucket, signs preserved
  accumulate tests/cuzk/smvp.rs:119-303        bucket sums == oracle_bucket_sums (SMVP sign folding, smvp.metal:46-105)
  reduce     tests/cuzk/pbpr.rs:26-247         bit sums == oracle_bit_sums; Horner over them == the MSM
on the two-level LDS sort, the tiled-histogram fallback, the global-atomic fallback and an oversized region (k_big_place); and on
the resident path's WINDOW TABLE (row f4: bucket arrays shared by the windows of a group, table indices in `sorted`, 2^19-bucket
arrays sorted with super-tiles and reduced as 2^16-bucket slices)."""
import numpy as np
import pytest

import mopro_msm_hip as mh
from mopro_msm_hip import testhooks as th
from conftest import load_golden
from oracle import bn254_oracle as orc

pytestmark = pytest.mark.gpu
SKIP, SIGN = 0xFFFFFFFF, 0x80000000


def signed_digits(scalars, c, W, signed=True):
    """the engine's recoding (csrc/msm_kernels.hpp k_decompose): v = window + carry; v > H => digit v - 2H, carry 1.
    (The reference recodes v >= H, convert_point...metal:97-116: same value, other tie rule -- Appendix B of SURVEY.md.)"""
    n = scalars.shape[0]
    vals = [orc.words_to_int(s) for s in scalars]
    H = 1 << (c - 1)
    out = np.zeros((W, n), np.int64)
    for i, v in enumerate(vals):
        carry = 0
        for w in range(W):
            d = ((v >> (c * w)) & ((1 << c) - 1)) + carry
            carry = 0
            if signed and d > H:
                d -= 2 * H
                carry = 1
            out[w, i] = d
        assert carry == 0
    return out


def placed(d, digits_signed):
    """where the engine PUTS a digit: bucket index + 1 with the digit's sign.  Every window but the top one: the digit itself.  The top
    window holds 254 - c*(W-1) bits (msm_plan_t.top_digit_bits = t): its bucket index is (|digit| - 1) | (point index mod 2^(kb-t)) << t, so
    that its entries use all of the window's buckets (csrc/msm_planner.hpp, k_decompose); the host leaves the bit sums u >= t out."""
    t, kb = d.plan.top_digit_bits, d.kb
    if t >= kb:
        return digits_signed
    out = digits_signed.copy()
    top = out[d.W - 1]
    mag = np.abs(top)
    assert (mag <= 1 << t).all(), "top window magnitude beyond 2^top_digit_bits"
    spread = (np.arange(top.shape[0]) & ((1 << (kb - t)) - 1)) << t
    out[d.W - 1] = np.where(mag > 0, np.sign(top) * (((mag - 1) | spread) + 1), 0)
    return out


def check_sort(d, digits_signed, inf=None):
    """offsets / sorted against the digits (transpose.rs:95-118: stable counting sort; here grouped, order inside a bucket free)"""
    digits_signed = placed(d, digits_signed)
    W, nb, nv = d.W, d.nb, d.nv
    assert d.offsets[0] == 0
    total = 0
    for w in range(W):
        row = digits_signed[w].copy()
        if inf is not None:
            row[inf != 0] = 0
        mag = np.abs(row)
        hist = np.bincount(mag[mag > 0] - 1, minlength=nb)
        off = d.offsets[w * nb: (w + 1) * nb + 1].astype(np.int64)
        assert (np.diff(off) == hist).all(), ("offsets != exclusive prefix of the digit histogram", w)
        # the device's digit codes: bucket | negate << 31, SKIP for zero digits / infinity
        code = np.where(mag > 0, (mag - 1).astype(np.uint32) | np.where(row < 0, SIGN, 0).astype(np.uint32), SKIP).astype(np.uint32)
        assert (d.digits[w] == code).all(), ("digit codes", w)
        seg = d.sorted[off[0]: off[-1]]
        idx = (seg & ~np.uint32(SIGN)).astype(np.int64)
        # a permutation of the non-skipped points of this window ...
        assert np.array_equal(np.sort(idx), np.flatnonzero(mag > 0)), ("sorted is not a permutation of the non-zero digits", w)
        # ... grouped by bucket, signs preserved
        bucket_of_slot = np.repeat(np.arange(nb), hist)
        assert (mag[idx] - 1 == bucket_of_slot).all(), ("entry in the wrong bucket", w)
        assert (((seg & SIGN) != 0) == (row[idx] < 0)).all(), ("sign lost", w)
        total += int((mag > 0).sum())
    assert int(d.offsets[W * nb]) == total  # the CSC end pointer == number of mixed additions k_accumulate will run


def same_point(a, b):
    xa, ia = orc.g1_to_affine_std(a)
    xb, ib = orc.g1_to_affine_std(b)
    return ia == ib and (xa == xb).all()


def check_buckets_and_bits(d, bases, form, digits_signed, inf, expected_affine):
    exp_b = orc.bucket_sums(bases, placed(d, digits_signed), d.nb, form, inf)
    for k in range(d.W * d.nb):
        assert same_point(d.buckets[k], exp_b[k]), ("bucket sum", k // d.nb, k % d.nb)
    exp_q = orc.bit_sums(exp_b, d.W, d.nb)
    for w in range(d.W):
        for u in range(d.kb + 1):
            assert same_point(d.bit_sums[w, u], exp_q[w, u]), ("bit sum", w, u)
    aff, inf_r = orc.g1_to_affine_std(d.jacobian)
    assert (aff == expected_affine).all()


CASES = [  # (window_bits, flags, golden case, what it exercises)
    (6, mh.FLAG_NO_GLV, "rand_n1024"),                            # two-level sort, tiny windows: long buckets, k_combine_long
    (10, mh.FLAG_NO_GLV, "rand_n4096"),                           # two-level sort
    (13, mh.FLAG_NO_GLV, "rand_n4096"),                           # mostly empty buckets
    (9, mh.FLAG_NO_GLV | mh.FLAG_UNSIGNED_DIGITS, "rand_n1024"),  # plain digits
    (8, mh.FLAG_NO_GLV, "edge_inf_bases"),
    (8, mh.FLAG_NO_GLV, "edge_same_base_same_scalar"),
    (7, mh.FLAG_NO_GLV, "edge_p_minus_p"),
]


@pytest.mark.parametrize("wb,flags,name", CASES)
def test_stages_two_level_sort(wb, flags, name):
    g = load_golden(name)
    with th.HooksContext(window_bits=wb, flags=flags) as c:
        d = c.stage_dump(g["bases"], g["scalars"], mh.FORM_STD, g["inf"])
    assert d.sort_path == 2
    ds = signed_digits(g["scalars"], wb, d.W, signed=not (flags & mh.FLAG_UNSIGNED_DIGITS))
    check_sort(d, ds, g["inf"])
    check_buckets_and_bits(d, g["bases"], orc.FORM_STD, ds, g["inf"], g["expected"])


def test_stages_tiled_fallback(monkeypatch):
    """MSM_HIP_DIRECT_SCATTER: per-tile LDS histograms + k_scan_* + k_tile_scatter (the path of n > 2^24 points)"""
    monkeypatch.setenv("MSM_HIP_DIRECT_SCATTER", "1")
    g = load_golden("rand_n4096")
    with th.HooksContext(window_bits=11, flags=mh.FLAG_NO_GLV) as c:
        d = c.stage_dump(g["bases"], g["scalars"], mh.FORM_STD, g["inf"])
    assert d.sort_path == 1
    ds = signed_digits(g["scalars"], 11, d.W)
    check_sort(d, ds, g["inf"])
    check_buckets_and_bits(d, g["bases"], orc.FORM_STD, ds, g["inf"], g["expected"])


def test_stages_wide_windows_two_level():
    """c = 19: 2^18 buckets per window.  Round 3: the two-level sort covers them (10 coarse + 8 fine bits; 9 fine bits at c = 20) and
    the reduction takes each array as 2^16-bucket slices (pseudo-windows) whose offsets the host adds back.
    (offsets and sorted only: 14 x 2^18 bucket records would be 340 MB of Jacobian words)"""
    g = load_golden("rand_n1024")
    for c_bits in (19, 20):
        with th.HooksContext(window_bits=c_bits, flags=mh.FLAG_NO_GLV) as c:
            d = c.stage_dump(g["bases"], g["scalars"], mh.FORM_STD, g["inf"], want_buckets=False)
        assert d.sort_path == 2 and d.pw_bits == c_bits - 17 and d.bit_sums.shape == (d.W << d.pw_bits, 17, 24)
        check_sort(d, signed_digits(g["scalars"], c_bits, d.W), g["inf"])
        aff, _ = orc.g1_to_affine_std(d.jacobian)
        assert (aff == g["expected"]).all()


def test_stages_global_atomic_fallback():
    """unsigned 20-bit windows: 2^20 buckets per window are beyond the LDS sort (10 coarse + 10 fine bits) -> device-scope atomics in
    k_decompose + k_scan_* + k_scatter.  (offsets and sorted only)"""
    g = load_golden("rand_n1024")
    fl = mh.FLAG_NO_GLV | mh.FLAG_UNSIGNED_DIGITS
    with th.HooksContext(window_bits=20, flags=fl) as c:
        d = c.stage_dump(g["bases"], g["scalars"], mh.FORM_STD, g["inf"], want_buckets=False)
    assert d.sort_path == 0
    check_sort(d, signed_digits(g["scalars"], 20, d.W, signed=False), g["inf"])
    aff, _ = orc.g1_to_affine_std(d.jacobian)
    assert (aff == g["expected"]).all()


def test_stages_glv_tie_digits_leave_the_16_bit_skip_code_free():
    """Round 5: digits travel as 16-bit codes (bucket 0..14, negate 15, 0xFFFF = zero digit) where a window has <= 2^15 buckets.  The code
    0xFFFF would also be "bucket 2^15 - 1, negated" = the digit -2^15 = -H.  A window value of exactly H can be written +H or -H (+ carry); the
    split recode folds the HALF's sign into every digit, so a negative half with a window at H used to give -H (2^-16 per digit).  The tie is
    now broken by the half's sign: final digits lie in [-(H-1), H] whatever the sign of the half.  Scalars built from halves whose every 16-bit
    window is H or H + 1 (both signs of both halves) -- k = k1 + lambda k2 mod r, the lattice-reduced split returns the halves it was built
    from -- through the stage dump at c = 16: digit range, reconstruction, and the MSM result against the oracle."""
    import json
    import os
    lam = int(json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "glv_constants.json")))["lambda"], 16)
    H = 1 << 15
    pats = [sum(H << (16 * j) for j in range(7)), sum((H + 1) << (16 * j) for j in range(7)), sum((H if j % 2 else H - 1) << (16 * j) for j in range(7)),
            H, H << 16, (H << 96) + H]
    ks = []
    for a in pats:
        for b in pats[:3]:
            for sa in (1, -1):
                for sb in (1, -1):
                    ks.append((sa * a + lam * sb * b) % R_ORDER)
    n = len(ks)
    scalars = np.stack([orc.int_to_words(k) for k in ks])
    logs = orc.gen_scalars(0x716, n, nonzero=True)
    bases = orc.gen_bases_from_logs(logs, orc.FORM_STD)
    with th.HooksContext(window_bits=16) as c:
        d = c.stage_dump(bases, scalars, mh.FORM_STD, None)
    assert d.plan.glv == 1 and d.plan.window_bits == 16 and d.kb == 15
    W, t = d.W, d.plan.top_digit_bits
    halves = np.zeros((2, n), object)
    ties = 0
    for w in range(W):
        code = d.digits[w].astype(np.int64)
        live = code != SKIP
        idx = code & 0x7FFFFFFF
        neg = (code & SIGN) != 0
        if w == W - 1 and t < d.kb:
            idx = idx & ((1 << t) - 1)
        mag = np.where(live, idx + 1, 0)
        assert (mag[neg & live] <= H - 1).all(), ("a negated magnitude of H: the 16-bit skip code", w)
        assert (mag <= H).all()
        ties += int((mag == H).sum())
        dig = np.where(neg, -mag, mag)
        for h in range(2):
            for i in range(n):
                halves[h, i] += int(dig[h * n + i]) << (16 * w)
    assert ties >= n, "the crafted halves must produce window values of exactly H"
    for i in range(n):
        assert (halves[0, i] + lam * halves[1, i] - ks[i]) % R_ORDER == 0, i
    exp, einf, _ = orc.msm_pippenger(bases, scalars, orc.FORM_STD)
    aff, inf = orc.g1_to_affine_std(d.jacobian)
    assert inf == int(einf) and (aff == exp).all()
    with mh.MsmContext() as c:  # the product library, default plan
        r = c.msm(bases, scalars, mh.FORM_STD)
        assert (r.affine_std == exp).all()


# ---- row f4: the window table of a resident base set, shared bucket arrays ---------------------------------------------------------
def check_sort_table(d, digits_signed, inf=None):
    """as check_sort, for V = W / f bucket arrays: array v sorts the f*nv digits of windows v*f .. v*f+f-1 and an entry is the TABLE
    index j*nv + i (window j of the group, point i)"""
    V, f, nb, nv = d.V, d.tf, d.nb, d.nv
    assert d.offsets[0] == 0 and V * f == d.W
    total = 0
    for v in range(V):
        rows = digits_signed[v * f:(v + 1) * f].copy()
        if inf is not None:
            rows[:, inf != 0] = 0
        row = rows.reshape(-1)
        mag = np.abs(row)
        hist = np.bincount(mag[mag > 0] - 1, minlength=nb)
        off = d.offsets[v * nb: (v + 1) * nb + 1].astype(np.int64)
        assert (np.diff(off) == hist).all(), ("offsets != exclusive prefix of the group's digit histogram", v)
        code = np.where(mag > 0, (mag - 1).astype(np.uint32) | np.where(row < 0, SIGN, 0).astype(np.uint32), SKIP).astype(np.uint32)
        assert (d.digits[v * f:(v + 1) * f].reshape(-1) == code).all(), ("digit codes", v)
        seg = d.sorted[off[0]: off[-1]]
        idx = (seg & ~np.uint32(SIGN)).astype(np.int64)
        assert np.array_equal(np.sort(idx), np.flatnonzero(mag > 0)), ("sorted is not a permutation of the group's non-zero digits", v)
        assert (mag[idx] - 1 == np.repeat(np.arange(nb), hist)).all(), ("entry in the wrong bucket", v)
        assert (((seg & SIGN) != 0) == (row[idx] < 0)).all(), ("sign lost", v)
        total += int((mag > 0).sum())
    assert int(d.offsets[V * nb]) == total


def top_shift(d, c_bits, scalar_bits=254):
    """ONE shared array (table factor == windows): the short top window's digit d enters as d * 2^s and its table level is
    2^(c*(W-1) - s) P (csrc/msm_hip.hip table_top_shift): s = (c - 1) - top bits"""
    if d.tf != d.W or d.tf <= 1:
        return 0
    top_bits = scalar_bits - c_bits * (d.W - 1)
    return max(0, c_bits - 1 - top_bits)


def table_digits(d, scalars, c_bits):
    ds = signed_digits(scalars, c_bits, d.W)
    ds[d.W - 1] <<= top_shift(d, c_bits)
    return ds


def table_bucket_sums(bases, form, digits_signed, inf, d, c_bits, buckets=None):
    """expected shared buckets: B[v][b] = sum_j 2^(c*j) * (bucket b of window v*f + j)  -- Horner over the group's windows
    (the top level of a full table is 2^(c*(W-1) - s) P: its first Horner step doubles s times fewer)"""
    per_window = orc.bucket_sums(bases, digits_signed, d.nb, form, inf).reshape(d.W, d.nb, 24)
    out = np.zeros((d.V * d.nb, 24), np.uint32)
    sh = top_shift(d, c_bits)
    for v in range(d.V):
        for b in (range(d.nb) if buckets is None else buckets):
            acc = per_window[v * d.tf + d.tf - 1, b]
            for j in range(d.tf - 2, -1, -1):
                acc = orc.g1_add(orc.g1_dbl_n(acc, c_bits - (sh if j == d.tf - 2 else 0)), per_window[v * d.tf + j, b])
            out[v * d.nb + b] = acc
    return out


TABLE_CASES = [  # (window_bits, table factor (0 = all windows), golden case)
    (8, 0, "rand_n1024"),            # ONE array shared by all 32 windows
    (8, 4, "rand_n1024"),            # 8 arrays of 4 windows each
    (6, 0, "edge_inf_bases"),
    (7, 0, "edge_same_base_same_scalar"),
    (10, 0, "edge_p_minus_p"),
    (13, 5, "rand_n4096"),
]


@pytest.mark.parametrize("wb,tf,name", TABLE_CASES)
def test_stages_window_table_shared_buckets(monkeypatch, wb, tf, name):
    """MSM_FLAG_WINDOW_TABLE, stage by stage: digits as without a table; offsets/sorted per bucket ARRAY with table indices; the
    shared buckets hold sum_j 2^(c*j) x (that bucket of window j); bit sums and the final point follow"""
    if tf:
        monkeypatch.setenv("MSM_HIP_TABLE_F", str(tf))
    g = load_golden(name)
    fl = mh.FLAG_NO_GLV | mh.FLAG_WINDOW_TABLE
    with th.HooksContext(window_bits=wb, flags=fl) as c:
        d = c.stage_dump(g["bases"], g["scalars"], mh.FORM_STD, g["inf"])
    assert d.tf == (tf or d.W) and d.V == d.W // d.tf and d.sort_path == 2
    ds = table_digits(d, g["scalars"], wb)
    check_sort_table(d, ds, g["inf"])
    exp_b = table_bucket_sums(g["bases"], orc.FORM_STD, ds, g["inf"], d, wb)
    for k in range(d.V * d.nb):
        assert same_point(d.buckets[k], exp_b[k]), ("shared bucket", k // d.nb, k % d.nb)
    exp_q = orc.bit_sums(exp_b, d.V, d.nb)
    for v in range(d.V):
        for u in range(d.kb + 1):
            assert same_point(d.bit_sums[v, u], exp_q[v, u]), ("bit sum", v, u)
    aff, _ = orc.g1_to_affine_std(d.jacobian)
    assert (aff == g["expected"]).all()


def test_stages_window_table_c20_super_tiles_and_pseudo_windows():
    """the shape the table exists for: c = 20, ONE array of 2^19 buckets shared by 13 windows.  The sort runs with 9 fine bits, so a
    staged element keeps 22 bits of its position and the rest is recovered from where it sits in its region (sort_hi; 4096 points x 13
    windows stay inside one super-tile, the 2^19-point test of tests/test_gpu_1_parity.py crosses them); the reduction sees 8 slices
    of 2^16 buckets.  Buckets are compared where the oracle has entries, the rest must be the identity."""
    g = load_golden("rand_n4096")
    fl = mh.FLAG_NO_GLV | mh.FLAG_WINDOW_TABLE
    with th.HooksContext(window_bits=20, flags=fl) as c:
        d = c.stage_dump(g["bases"], g["scalars"], mh.FORM_STD, g["inf"])
    assert (d.tf, d.V, d.W, d.kb, d.pw_bits, d.rkb) == (13, 1, 13, 19, 3, 16) and d.sort_path == 2
    assert top_shift(d, 20) == 5  # the 14-bit top window enters as d * 32
    ds = table_digits(d, g["scalars"], 20)
    check_sort_table(d, ds, g["inf"])
    mags = np.abs(ds)
    mags[:, g["inf"] != 0] = 0
    used = np.unique(mags[mags > 0] - 1)
    exp_b = table_bucket_sums(g["bases"], orc.FORM_STD, ds, g["inf"], d, 20, buckets=used)
    for b in used:
        assert same_point(d.buckets[b], exp_b[b]), ("shared bucket", int(b))
    empty = np.setdiff1d(np.arange(d.nb), used)
    assert (d.buckets[empty, 16:24] == 0).all()  # Z == 0: identity
    exp_b[empty] = d.buckets[empty]              # ... in the words the device wrote
    exp_q = orc.bit_sums(exp_b, d.V << d.pw_bits, 1 << d.rkb)
    for q in range(d.V << d.pw_bits):
        for u in range(d.rkb + 1):
            assert same_point(d.bit_sums[q, u], exp_q[q, u]), ("bit sum of slice", q, u)
    aff, _ = orc.g1_to_affine_std(d.jacobian)
    assert (aff == g["expected"]).all()


def test_stages_oversized_region_big_place():
    """skewed scalars: one bucket holds most of every window, its sort region exceeds a workgroup's staging area and is cut
    into batches that worker blocks count and k_big_place places (msm_kernels.hpp k_coarse_starts / k_big_place)"""
    n = 1 << 15
    k = orc.gen_scalars(31, n, nonzero=True)
    bases = orc.gen_bases_from_logs(k, orc.FORM_MONT)
    s = orc.gen_scalars(32, n)
    s[: n - n // 8] = s[0]           # 7/8 of the points share one scalar: every window has one region of ~28000 entries
    s[n // 2: n // 2 + 100, 1:] = 0  # and some tiny ones
    with th.HooksContext(window_bits=12, flags=mh.FLAG_NO_GLV) as c:
        d = c.stage_dump(bases, s, mh.FORM_MONT, None)
    assert d.sort_path == 2 and d.big_items > 0, "the oversized-region path was not taken"
    ds = signed_digits(s, 12, d.W)
    check_sort(d, ds)
    exp, einf = orc.closed_form_expected(k, s)
    check_buckets_and_bits(d, bases, orc.FORM_MONT, ds, None, exp)


def test_stages_moderately_oversized_region_sorted_by_its_owner():
    """a region between one and four staging areas long (here ~10000 entries against 4096) stays with its owner workgroup, which
    counts and places it batch after batch straight into `sorted` (k_fine_sort, staged == false) -- the path every region of a
    2^22-point window-table MSM takes; none is handed to the worker blocks"""
    n = 1 << 15
    k = orc.gen_scalars(41, n, nonzero=True)
    bases = orc.gen_bases_from_logs(k, orc.FORM_MONT)
    s = orc.gen_scalars(42, n)
    s[: n // 4] = s[0]  # a quarter of the points share one scalar: every window has one region of ~8192 + 96 entries
    # (c = 13: the 7-bit top window spreads over 8 regions of ~3000-11000 entries; at c = 12 it would be one region of 24576)
    with th.HooksContext(window_bits=13, flags=mh.FLAG_NO_GLV) as c:
        d = c.stage_dump(bases, s, mh.FORM_MONT, None)
    assert d.sort_path == 2 and d.big_items == 0
    ds = signed_digits(s, 13, d.W)
    check_sort(d, ds)
    exp, einf = orc.closed_form_expected(k, s)
    check_buckets_and_bits(d, bases, orc.FORM_MONT, ds, None, exp)


def test_stages_glv_plan_sort_invariants_and_result():
    """default plan at n = 4096 (GLV split: 2n virtual points, digits of the two 127-bit halves): the sort invariants hold on the
    dumped digit codes, the bit sums are the bit sums of the dumped buckets, Horner over them is the golden MSM"""
    g = load_golden("rand_n4096")
    with th.HooksContext() as c:
        d = c.stage_dump(g["bases"], g["scalars"], mh.FORM_STD, g["inf"])
    assert d.plan.glv == 1 and d.nv == 2 * 4096
    W, nb = d.W, d.nb
    for w in range(W):
        code = d.digits[w]
        live = code != SKIP
        mag = np.where(live, (code & ~np.uint32(SIGN)).astype(np.int64) + 1, 0)
        hist = np.bincount(mag[mag > 0] - 1, minlength=nb)
        off = d.offsets[w * nb: (w + 1) * nb + 1].astype(np.int64)
        assert (np.diff(off) == hist).all()
        seg = d.sorted[off[0]: off[-1]]
        idx = (seg & ~np.uint32(SIGN)).astype(np.int64)
        assert np.array_equal(np.sort(idx), np.flatnonzero(live))
        assert (mag[idx] - 1 == np.repeat(np.arange(nb), hist)).all()
        assert ((seg & SIGN) == (code[idx] & SIGN)).all()
    exp_q = orc.bit_sums(d.buckets, W, nb)
    for w in range(W):
        for u in range(d.kb + 1):
            assert same_point(d.bit_sums[w, u], exp_q[w, u])
    aff, _ = orc.g1_to_affine_std(d.jacobian)
    assert (aff == g["expected"]).all()


R_ORDER = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001


@pytest.mark.parametrize("wb,name", [(0, "rand_n4096"), (16, "rand_n1024"), (13, "rand_n1024"), (9, "rand_n1024")])
def test_stages_glv_digits_rebuild_the_halves_and_the_top_window_is_spread(wb, name):
    """split plans, decompose stage: the digit codes of virtual points i and n + i rebuild k1 and k2 with k1 + lambda k2 = k (mod r) and
    |k_j| < 7 * 2^123; in the TOP window the bucket index carries the magnitude in its low msm_plan_t.top_digit_bits bits and the low bits
    of the point index above them (csrc/msm_planner.hpp glv_top_digit_bits: its buckets as full as the other windows')."""
    import json
    import os
    lam = int(json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "glv_constants.json")))["lambda"], 16)
    g = load_golden(name)
    n = g["scalars"].shape[0]
    with th.HooksContext(window_bits=wb) as c:
        d = c.stage_dump(g["bases"], g["scalars"], mh.FORM_STD, g["inf"])
    pl = d.plan
    assert pl.glv == 1 and d.nv == 2 * n
    W, kb, t, cb = d.W, d.kb, pl.top_digit_bits, pl.window_bits
    # the planner's rule, restated: the halves are below 7 * 2^123
    maxmag = (((7 << 123) - 1) >> (cb * (W - 1))) + 1
    assert t == min(kb, max(0, (maxmag - 1).bit_length())) and maxmag <= 1 << t
    live_pts = np.ones(n, bool) if g["inf"] is None else (np.asarray(g["inf"]) == 0)
    halves = np.zeros((2, n), object)
    for w in range(W):
        code = d.digits[w].astype(np.int64)
        live = code != SKIP
        idx = code & 0x7FFFFFFF
        neg = (code & SIGN) != 0
        if w == W - 1 and t < kb:
            pt = np.tile(np.arange(n), 2)
            assert ((idx >> t)[live] == (pt & ((1 << (kb - t)) - 1))[live]).all(), "spread bits are the low bits of the point index"
            idx = idx & ((1 << t) - 1)
            if live.sum() >= 16 << (kb - t):  # the spread reaches the top of the window's index range
                assert (code[live] & 0x7FFFFFFF).max() >> t == (1 << (kb - t)) - 1
        mag = np.where(live, idx + 1, 0)
        dig = np.where(neg, -mag, mag)
        for h in range(2):
            for i in range(n):
                halves[h, i] += int(dig[h * n + i]) << (cb * w)
    for i in range(n):
        if not live_pts[i]:
            assert halves[0, i] == 0 and halves[1, i] == 0
            continue
        k = orc.words_to_int(g["scalars"][i])
        assert (halves[0, i] + lam * halves[1, i] - k) % R_ORDER == 0, i
        assert abs(halves[0, i]) < 7 << 123 and abs(halves[1, i]) < 7 << 123
    aff, _ = orc.g1_to_affine_std(d.jacobian)
    assert (aff == g["expected"]).all()
