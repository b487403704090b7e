"""Test, benchmark and calibration hooks (include/msm_hip_testhooks.h) -- NOT part of the product.

They live in a second build of the same sources, gpu-acceleration_amd/libmsm_hip_hooks.so (-DMSM_HIP_TEST_HOOKS), which also
contains the whole engine: a HooksContext is an MsmContext on that build plus the hooks.  Users: tests/, bench.py (input
generation, multiplier calibration), tools/.  The product library libmsm_hip.so exports none of these symbols.
"""
import ctypes as C
import os

import numpy as np

from . import (ERR_NO_DEVICE, FLAG_NO_GLV, MsmContext, MsmError, _p32, _preload_torch, _u8p, _u32p, _words, bind_product_abi, plan)

_HERE = os.path.dirname(os.path.abspath(__file__))
HOOKS_LIB_PATH = os.environ.get("MSM_HIP_HOOKS_LIB") or os.path.join(os.path.dirname(_HERE), "libmsm_hip_hooks.so")

HOOK_SYMBOLS = [
    "msm_bn254_g1_generate_device", "msm_bn254_generate_scalars_host", "msm_test_fp_op", "msm_test_g1_op",
    "msm_test_decompose", "msm_calibrate", "msm_test_stage_dump", "msm_test_abandon_after_sort",
    "msm_probe_wide_level", "msm_probe_launch_chain", "msm_probe_empty_launch", "msm_test_get_list_counts", "msm_probe_reduce_bits",
]
_lib = None


def hooks_plan(n, window_bits=0, flags=0):
    """msm_plan of the HOOKS build: the one that reads the test / A-B knobs (MSM_HIP_GLV_MAX_LOG2, MSM_HIP_TABLE_* ...); the product's reads none"""
    return plan(n, window_bits, flags, _lib=load_hooks_library())


def load_hooks_library():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(HOOKS_LIB_PATH):
        raise MsmError(ERR_NO_DEVICE, f"{HOOKS_LIB_PATH} not built: make -C gpu-acceleration_amd/csrc hooks")
    _preload_torch()
    L = bind_product_abi(C.CDLL(HOOKS_LIB_PATH))
    vp = C.c_void_p
    L.msm_bn254_g1_generate_device.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_size_t, vp, vp]
    L.msm_bn254_generate_scalars_host.argtypes = [C.c_uint64, C.c_size_t, C.c_int, _u32p]
    L.msm_test_fp_op.argtypes = [vp, C.c_uint32, _u32p, _u32p, _u32p, C.c_size_t]
    L.msm_test_g1_op.argtypes = [vp, C.c_uint32, _u32p, _u32p, _u32p, C.c_size_t]
    L.msm_test_decompose.argtypes = [vp, _u32p, C.c_size_t, C.c_uint32, C.POINTER(C.c_int32)]
    L.msm_test_abandon_after_sort.argtypes = [vp, _u32p, C.c_size_t]
    L.msm_calibrate.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.msm_probe_wide_level.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_longlong)]
    L.msm_probe_launch_chain.argtypes = [vp, C.c_uint32, C.c_uint32, C.POINTER(C.c_double)]
    L.msm_probe_empty_launch.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_double)]
    L.msm_test_get_list_counts.argtypes = [vp, _u32p]
    L.msm_probe_reduce_bits.argtypes = [vp, C.c_uint32, C.c_uint32, C.POINTER(C.c_double)]
    L.msm_test_stage_dump.argtypes = [vp, _u32p, C.c_uint32, _u8p, _u32p, C.c_size_t, _u32p, _u32p, _u32p, _u32p, _u32p,
                                      C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), _u32p]
    for name in HOOK_SYMBOLS:
        getattr(L, name).restype = C.c_int32
    _lib = L
    return L


def generate_scalars_host(seed, n, nonzero=False):
    """element i of SplitMix64 stream `seed` reduced below r -- the k_i / s_i of the synthetic instances, on the host"""
    out = np.zeros((n, 8), np.uint32)
    load_hooks_library().msm_bn254_generate_scalars_host(seed, n, int(nonzero), _p32(out))
    return out


class StageDump:
    """intermediates of one pipeline run (msm_test_stage_dump)"""


class HooksContext(MsmContext):
    """MsmContext on the hooks build + the hooks themselves"""
    _loader = staticmethod(lambda: load_hooks_library())

    def generate_device(self, base_seed, scalar_seed, n, d_bases_ptr, d_scalars_ptr):
        """synthetic instance straight into device memory: bases k_i*G (Montgomery words), scalars s_i (counterpart of
        test_utils::generate_random_bases_and_scalars, metal_msm.rs:698-731)"""
        self._check(self._lib.msm_bn254_g1_generate_device(self._h, base_seed, scalar_seed, n, d_bases_ptr, d_scalars_ptr))

    # -- device-math unit-test hooks -----------------------------------------------------------
    def test_fp_op(self, op, a, b=None):
        a = _words(a, 8)
        b = _words(b, 8) if b is not None else None
        out = np.zeros_like(a)
        self._check(self._lib.msm_test_fp_op(self._h, op, _p32(a), _p32(b), _p32(out), a.shape[0]))
        return out

    def test_g1_op(self, op, a, b=None):
        a = _words(a, 24)
        b = _words(b, 16 if op in (0, 4, 5) else 24) if b is not None else None  # (0, 4, 5: mixed additions, b affine)
        out = np.zeros_like(a)
        self._check(self._lib.msm_test_g1_op(self._h, op, _p32(a), _p32(b), _p32(out), a.shape[0]))
        return out

    def abandon_after_sort(self, scalars):
        """decomposition + sort + piece plan of an MSM on `scalars`, then the failure a copy / event wait in front of the accumulation would be:
        raises MsmError(ERR_HIP); the context is left as that failure leaves it"""
        scalars = _words(scalars, 8)
        self._check(self._lib.msm_test_abandon_after_sort(self._h, _p32(scalars), scalars.shape[0]))

    def calibrate(self):
        """(v_mad_u64_u32 per second, field multiplications per second) this device sustains -- two ~1 ms micro-kernels."""
        a, b = C.c_double(0), C.c_double(0)
        self._check(self._lib.msm_calibrate(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def probe_wide_level(self, threads, m, iters, mode):
        """one workgroup, `iters` pairwise levels over m records in LDS (msm_probe_wide_level): 18 counters, see include/msm_hip_testhooks.h"""
        out = (C.c_longlong * 18)()
        self._check(self._lib.msm_probe_wide_level(self._h, threads, m, iters, mode, out))
        return [int(v) for v in out]

    def probe_empty_launch(self, blocks, threads, lds_kb, launches=200):
        """microseconds per dependent launch of an empty kernel of blocks x threads with lds_kb of static LDS per workgroup"""
        us = C.c_double(0)
        self._check(self._lib.msm_probe_empty_launch(self._h, blocks, threads, lds_kb, launches, C.byref(us)))
        return us.value

    def probe_reduce_bits(self, parts, launches=100):
        """microseconds per launch of k_reduce_bits_wide<parts> on the 2^20 shape (bit 0 staging, 1 tree, 2 final conversion)"""
        us = C.c_double(0)
        self._check(self._lib.msm_probe_reduce_bits(self._h, parts, launches, C.byref(us)))
        return us.value

    def list_counts(self):
        """{"long", "mid", "pieces", "partials", "mid2"}: the list counters the last sort chain / accumulation left on the device"""
        out = np.zeros(5, np.uint32)
        self._check(self._lib.msm_test_get_list_counts(self._h, _p32(out)))
        return dict(zip(("long", "mid", "pieces", "partials", "mid2"), (int(v) for v in out)))

    def probe_launch_chain(self, n_adds, launches):
        """microseconds per dependent launch of k_pair_level_wide over n_adds additions (0: an empty kernel)"""
        us = C.c_double(0)
        self._check(self._lib.msm_probe_launch_chain(self._h, n_adds, launches, C.byref(us)))
        return us.value

    def test_decompose(self, scalars, window_bits=0):
        scalars = _words(scalars, 8)
        n = scalars.shape[0]
        p = hooks_plan(n, window_bits or self.window_bits, self.flags | FLAG_NO_GLV)  # the hook returns the plain 254-bit digits
        out = np.zeros((p.num_windows, n), np.int32)
        self._check(self._lib.msm_test_decompose(self._h, _p32(scalars), n, window_bits, out.ctypes.data_as(C.POINTER(C.c_int32))))
        return out

    def stage_dump(self, bases, scalars, form=0, inf=None, want_buckets=True):
        """run the pipeline once and return every stage's intermediates (counterpart of the reference's per-kernel tests)"""
        bases, scalars = _words(bases, 16), _words(scalars, 8)
        n = min(bases.shape[0], scalars.shape[0])
        p = hooks_plan(n, self.window_bits, self.flags)
        W, nb, nv = p.num_windows, p.num_buckets, int(p.virtual_points)
        kb = nb.bit_length() - 1
        d = StageDump()
        d.plan, d.W, d.nb, d.nv, d.kb = p, W, nb, nv, kb
        # bucket arrays (= windows without a window table), table factor, pseudo-windows of the reduction (arrays above 2^17 buckets)
        d.V, d.tf = int(p.bucket_arrays), int(p.table_factor)
        d.pw_bits = kb - 16 if kb > 17 else 0
        d.rkb = kb - d.pw_bits
        d.digits = np.zeros((W, nv), np.uint32)
        d.offsets = np.zeros(d.V * nb + 1, np.uint32)
        d.sorted = np.zeros(W * nv, np.uint32)
        d.buckets = np.zeros((d.V * nb, 24), np.uint32) if want_buckets else None
        d.bit_sums = np.zeros((d.V << d.pw_bits, d.rkb + 1, 24), np.uint32)
        d.jacobian = np.zeros(24, np.uint32)
        sp, bi = C.c_uint32(0), C.c_uint32(0)
        infp = None
        if inf is not None:
            inf = np.ascontiguousarray(inf, dtype=np.uint8)
            infp = inf.ctypes.data_as(_u8p)
        self._check(self._lib.msm_test_stage_dump(self._h, _p32(bases), form, infp, _p32(scalars), n, _p32(d.digits), _p32(d.offsets),
                                                  _p32(d.sorted), _p32(d.buckets) if want_buckets else None, _p32(d.bit_sums),
                                                  C.byref(sp), C.byref(bi), _p32(d.jacobian)))
        d.sort_path, d.big_items = int(sp.value), int(bi.value)
        return d
