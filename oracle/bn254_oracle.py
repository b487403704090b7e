"""ctypes binding of oracle/liboracle_bn254.so (the CPU restatement, bn254_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_DIR, "liboracle_bn254.so")

P = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
R_ORDER = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
FORM_STD, FORM_MONT = 0, 1

_u32p = C.POINTER(C.c_uint32)
_u8p = C.POINTER(C.c_uint8)
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _DIR])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        msm_args = [_u32p, C.c_uint32, _u8p, _u32p, C.c_size_t]
        L.oracle_msm_naive.argtypes = msm_args + [_u32p, _u8p, _u32p]
        L.oracle_msm_pippenger.argtypes = msm_args + [C.c_int, _u32p, _u8p, _u32p]
        L.oracle_msm_cuzk.argtypes = msm_args + [C.c_uint32, _u32p, _u8p, _u32p]
        for f in (L.oracle_msm_naive, L.oracle_msm_pippenger, L.oracle_msm_cuzk):
            f.restype = C.c_int
        L.oracle_ref_window_bits.argtypes = [C.c_size_t]
        L.oracle_ref_window_bits.restype = C.c_uint32
        L.oracle_num_windows.argtypes = [C.c_uint32]
        L.oracle_num_windows.restype = C.c_uint32
        L.oracle_decompose_signed.argtypes = [_u32p, C.c_size_t, C.c_uint32, _u32p]
        L.oracle_transpose.argtypes = [_u32p, C.c_size_t, C.c_uint32, C.c_uint32, _u32p, _u32p]
        L.oracle_bucket_sums.argtypes = [_u32p, C.c_uint32, _u8p, C.POINTER(C.c_int32), C.c_size_t, C.c_uint32, C.c_uint32, _u32p]
        L.oracle_bucket_sums.restype = C.c_int
        L.oracle_bit_sums.argtypes = [_u32p, C.c_uint32, C.c_uint32, C.c_uint32, _u32p]
        L.oracle_bit_sums.restype = C.c_int
        L.oracle_gen_scalars.argtypes = [C.c_uint64, C.c_size_t, C.c_int, _u32p]
        L.oracle_gen_bases_from_logs.argtypes = [_u32p, C.c_size_t, C.c_uint32, _u32p]
        L.oracle_g1_to_affine_std.restype = C.c_int
        L.oracle_g1_dbl_n.argtypes = [_u32p, C.c_uint32, _u32p]
        L.oracle_g1_decompress.argtypes = [_u8p, C.c_size_t, C.c_uint32, _u32p, _u8p]
        L.oracle_g1_decompress.restype = C.c_size_t
        L.oracle_g1_compress.argtypes = [_u32p, C.c_uint32, _u8p, C.c_size_t, _u8p]
        L.oracle_threads_available.restype = C.c_int
        _lib = L
    return _lib


def _p32(a):
    return a.ctypes.data_as(_u32p)


def _w(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.uint32)
    if shape is not None:
        a = a.reshape(shape)
    return a


def int_to_words(v, n=8):
    return np.array([(v >> (32 * i)) & 0xFFFFFFFF for i in range(n)], dtype=np.uint32)


def words_to_int(ws):
    v = 0
    for i, w in enumerate(np.asarray(ws).reshape(-1).tolist()):
        v |= int(w) << (32 * i)
    return v


# ---- field --------------------------------------------------------------------
def fq_constants():
    p, r1, r2 = (np.zeros(8, np.uint32) for _ in range(3))
    inv = C.c_uint64(0)
    lib().oracle_fq_constants(_p32(p), _p32(r1), _p32(r2), C.byref(inv))
    return words_to_int(p), words_to_int(r1), words_to_int(r2), inv.value


def _un(fn, a):
    a = _w(a)
    o = np.zeros(8, np.uint32)
    fn(_p32(a), _p32(o))
    return o


def _bin(fn, a, b):
    a, b = _w(a), _w(b)
    o = np.zeros(8, np.uint32)
    fn(_p32(a), _p32(b), _p32(o))
    return o


def fq_to_mont(a): return _un(lib().oracle_fq_to_mont, a)
def fq_from_mont(a): return _un(lib().oracle_fq_from_mont, a)
def fq_inv_mont(a): return _un(lib().oracle_fq_inv_mont, a)
def fq_mont_mul(a, b): return _bin(lib().oracle_fq_mont_mul, a, b)
def fq_add(a, b): return _bin(lib().oracle_fq_add, a, b)
def fq_sub(a, b): return _bin(lib().oracle_fq_sub, a, b)


# ---- group ---------------------------------------------------------------------
def g1_dbl(a):
    a = _w(a); o = np.zeros(24, np.uint32); lib().oracle_g1_dbl(_p32(a), _p32(o)); return o


def g1_dbl_n(a, k):
    """2^k * a"""
    a = _w(a); o = np.zeros(24, np.uint32); lib().oracle_g1_dbl_n(_p32(a), int(k), _p32(o)); return o


def g1_add(a, b):
    a, b = _w(a), _w(b); o = np.zeros(24, np.uint32); lib().oracle_g1_add(_p32(a), _p32(b), _p32(o)); return o


def g1_madd(a, b_xy_mont):
    a, b = _w(a), _w(b_xy_mont); o = np.zeros(24, np.uint32); lib().oracle_g1_madd(_p32(a), _p32(b), _p32(o)); return o


def g1_to_affine_std(a):
    a = _w(a); o = np.zeros(16, np.uint32)
    inf = lib().oracle_g1_to_affine_std(_p32(a), _p32(o))
    return o, int(inf)


def g1_scalar_mul(base_xy_std, k):
    b, k = _w(base_xy_std), _w(k); o = np.zeros(24, np.uint32)
    lib().oracle_g1_scalar_mul(_p32(b), _p32(k), _p32(o)); return o


# ---- MSM -----------------------------------------------------------------------
def _msm(fn, bases, scalars, form, inf, *extra):
    bases = _w(bases).reshape(-1, 16)
    scalars = _w(scalars).reshape(-1, 8)
    n = min(bases.shape[0], scalars.shape[0])
    infp = None
    if inf is not None:
        inf = np.ascontiguousarray(inf, dtype=np.uint8)
        infp = inf.ctypes.data_as(_u8p)
    out = np.zeros(16, np.uint32)
    jac = np.zeros(24, np.uint32)
    oi = C.c_uint8(0)
    rc = fn(_p32(bases), form, infp, _p32(scalars), n, *extra, _p32(out), C.byref(oi), _p32(jac))
    if rc != 0:
        raise RuntimeError(f"oracle MSM failed rc={rc}")
    return out, int(oi.value), jac


def msm_naive(bases, scalars, form=FORM_STD, inf=None):
    return _msm(lib().oracle_msm_naive, bases, scalars, form, inf)


def msm_pippenger(bases, scalars, form=FORM_STD, inf=None, threads=0):
    return _msm(lib().oracle_msm_pippenger, bases, scalars, form, inf, C.c_int(threads))


def msm_cuzk(bases, scalars, form=FORM_STD, inf=None, window_bits=0):
    return _msm(lib().oracle_msm_cuzk, bases, scalars, form, inf, C.c_uint32(window_bits))


def threads_available():
    return int(lib().oracle_threads_available())


# ---- reference stage mirrors ---------------------------------------------------
def ref_window_bits(n): return int(lib().oracle_ref_window_bits(n))
def num_windows(w): return int(lib().oracle_num_windows(w))


def decompose_signed(scalars, window_bits):
    scalars = _w(scalars).reshape(-1, 8)
    n = scalars.shape[0]
    W = num_windows(window_bits)
    out = np.zeros((W, n), np.uint32)
    lib().oracle_decompose_signed(_p32(scalars), n, window_bits, _p32(out))
    return out


def transpose(chunks, num_cols):
    chunks = _w(chunks)
    W, n = chunks.shape
    col_ptr = np.zeros((W, num_cols + 1), np.uint32)
    val = np.zeros((W, n), np.uint32)
    lib().oracle_transpose(_p32(chunks), n, W, num_cols, _p32(col_ptr), _p32(val))
    return col_ptr, val


def bucket_sums(bases, digits, nb, form=FORM_STD, inf=None):
    """digits: (W, n) signed ints -> (W*nb, 24) Jacobian bucket sums (bucket b = magnitude b + 1)"""
    bases = _w(bases).reshape(-1, 16)
    d = np.ascontiguousarray(digits, dtype=np.int32)
    W, n = d.shape
    out = np.zeros((W * nb, 24), np.uint32)
    ip = None
    if inf is not None:
        inf = np.ascontiguousarray(inf, dtype=np.uint8)
        ip = inf.ctypes.data_as(_u8p)
    rc = lib().oracle_bucket_sums(_p32(bases), form, ip, d.ctypes.data_as(C.POINTER(C.c_int32)), n, W, nb, _p32(out))
    if rc != 0:
        raise RuntimeError(f"oracle_bucket_sums failed rc={rc}")
    return out


def bit_sums(buckets_jac, W, nb):
    kb = nb.bit_length() - 1
    b = _w(buckets_jac).reshape(W * nb, 24)
    out = np.zeros((W, kb + 1, 24), np.uint32)
    rc = lib().oracle_bit_sums(_p32(b), W, nb, kb, _p32(out))
    if rc != 0:
        raise RuntimeError(f"oracle_bit_sums failed rc={rc}")
    return out


# ---- synthetic inputs ----------------------------------------------------------
def gen_scalars(seed, n, nonzero=False):
    out = np.zeros((n, 8), np.uint32)
    lib().oracle_gen_scalars(seed, n, int(nonzero), _p32(out))
    return out


def gen_bases_from_logs(k, form=FORM_MONT):
    k = _w(k).reshape(-1, 8)
    out = np.zeros((k.shape[0], 16), np.uint32)
    lib().oracle_gen_bases_from_logs(_p32(k), k.shape[0], form, _p32(out))
    return out


def g1_decompress(compressed, form=FORM_MONT):
    """arkworks-0.4 compressed images (n x 32 bytes) -> (xy words n x 16, inf n, first_invalid or -1)."""
    buf = np.ascontiguousarray(np.frombuffer(bytes(compressed), dtype=np.uint8))
    n = buf.size // 32
    out = np.zeros((n, 16), np.uint32)
    inf = np.zeros(n, np.uint8)
    bad = lib().oracle_g1_decompress(buf.ctypes.data_as(_u8p), n, form, _p32(out), inf.ctypes.data_as(_u8p))
    return out, inf, int(bad) - 1


def g1_compress(bases, form=FORM_MONT, inf=None):
    b = _w(bases).reshape(-1, 16)
    out = np.zeros(b.shape[0] * 32, np.uint8)
    ip = None
    if inf is not None:
        inf = np.ascontiguousarray(inf, dtype=np.uint8)
        ip = inf.ctypes.data_as(_u8p)
    lib().oracle_g1_compress(_p32(b), form, ip, b.shape[0], out.ctypes.data_as(_u8p))
    return out.tobytes()


def dot_words(k_words, s_words):
    """sum_i k_i * s_i as a Python integer for two (n, 8) arrays of little-endian u32 words, vectorised: the words are cut into 16-bit
    halves and the 16 x 16 sums of half-products are taken by numpy in uint64 (each < 2^32 * n: exact up to 2^31 rows per call, done in
    slices of 2^20) -- 2^24 points in seconds instead of minutes of Python big-integer loops."""
    k = _w(k_words).reshape(-1, 8)
    s = _w(s_words).reshape(-1, 8)
    n = min(len(k), len(s))
    tot = 0
    for lo in range(0, n, 1 << 20):
        k16 = np.ascontiguousarray(k[lo:lo + (1 << 20)]).view(np.uint16).reshape(-1, 16).astype(np.float64)
        s16 = np.ascontiguousarray(s[lo:lo + (1 << 20)]).view(np.uint16).reshape(-1, 16).astype(np.float64)
        # float64 BLAS is exact here: every partial sum stays below 2^32 * 2^20 = 2^52 < 2^53
        m = k16.T @ s16
        for a in range(16):
            for b in range(16):
                tot += int(m[a, b]) << (16 * (a + b))
    return tot


def closed_form_expected(k_words, s_words):
    """(sum s_i*k_i mod r) * G as canonical affine standard words -- O(n) integer work."""
    k = _w(k_words).reshape(-1, 8).astype(object)
    s = _w(s_words).reshape(-1, 8).astype(object)
    n = min(len(k), len(s))
    tot = 0
    for i in range(n):
        ki = 0
        si = 0
        for j in range(7, -1, -1):
            ki = (ki << 32) | int(k[i, j])
            si = (si << 32) | int(s[i, j])
        tot += ki * si
    tot %= R_ORDER
    g = np.zeros(16, np.uint32)
    g[0] = 1
    g[8] = 2
    return g1_to_affine_std(g1_scalar_mul(g, int_to_words(tot)))
