"""mopro_msm_hip -- Python host-side mirror of the reference's MSM operator interface, over the
C ABI of libmsm_hip.so (include/msm_hip.h).

Reference surface mirrored (mopro-msm/src/msm/metal_msm/metal_msm.rs):
  metal_variable_base_msm(&bases, &scalars) -> Result<G1Projective, Box<dyn Error>>   :642-695
      * empty input            -> Err("Empty input")                                   :647-649
      * unequal lengths        -> silently truncated to the shorter one                :652-656
  test_utils::generate_random_bases_and_scalars(size)                                  :698-731
The Rust shim a maintainer would add is in rust/mopro-msm-hip; this module is the
same thin layer for Python callers, tests and bench.py.  It holds no arithmetic: every result comes
from the HIP library, and importing/using it without the built library or without a GPU raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# MSM_HIP_LIB overrides the in-tree library (A/B runs of two builds: tools/sweep_env.py MSM_HIP_LIB a.so b.so)
LIB_PATH = os.environ.get("MSM_HIP_LIB") or os.path.join(os.path.dirname(_HERE), "libmsm_hip.so")

FORM_STD, FORM_MONT = 0, 1
FLAG_UNSIGNED_DIGITS = 1
FLAG_NO_GLV = 2
FLAG_WINDOW_TABLE = 4  # resident sets carry their window table
FLAG_DETERMINISTIC = 8  # jacobian_mont is the canonical Z = 1 representative: the same 24 words for the same group element (ABI 6)
OK, ERR_EMPTY, ERR_BAD_ARG, ERR_NO_DEVICE, ERR_HIP, ERR_OOM, ERR_STATE, ERR_INVALID_DATA = 0, -1, -2, -3, -4, -5, -6, -7

# every symbol include/msm_hip.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = [
    "msm_abi_version", "msm_ctx_create", "msm_ctx_destroy", "msm_last_error", "msm_bn254_g1",
    "msm_bn254_g1_arkworks", "msm_bn254_g1_upload_bases", "msm_bn254_g1_resident", "msm_bn254_g1_resident_batch", "msm_tune_batch", "msm_bn254_g1_resident_device", "msm_bn254_g1_device",
    "msm_bn254_g1_combine", "msm_bn254_g1_combine_flags", "msm_get_timings_sized", "msm_multi_get_timings_sized", "msm_multi_get_clock_stats", "msm_multi_set_kernel_timing",
    "msm_plan", "msm_get_timings", "msm_set_stage_timing", "msm_set_kernel_timing", "msm_get_accumulate_kernel_stats", "msm_reset_kernel_stats", "msm_get_clock_stats",
    "msm_bn254_g1_decompress", "msm_bn254_g1_upload_compressed", "msm_bn254_g1_compress",
    "msm_multi_create", "msm_multi_destroy", "msm_multi_last_error", "msm_multi_num_devices", "msm_multi_exchange",
    "msm_bn254_g1_multi", "msm_bn254_g1_multi_arkworks", "msm_bn254_g1_multi_device", "msm_multi_get_timings",
    "msm_multi_get_exchange_stats", "msm_multi_get_exchange_probe",
]
ABI_VERSION = 7  # == MSM_HIP_ABI_VERSION of include/msm_hip.h this binding was written against (checked when a library is loaded)
ERR_RCCL = -8
EXCHANGE_AUTO, EXCHANGE_RCCL, EXCHANGE_HOST = 0, 1, 2
# msm_config_t.batch_layout (include/msm_hip.h MSM_BATCH_LAYOUT_*)
BATCH_LAYOUT_AUTO, BATCH_LAYOUT_ONE_STREAM, BATCH_LAYOUT_ONE_STREAM_REDUCE, BATCH_LAYOUT_TWO_STREAMS = 0, 1, 2, 3


class MsmError(RuntimeError):
    """Counterpart of the reference's Box<dyn Error>; .code is the C-ABI status."""

    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code


class Config(C.Structure):
    _fields_ = [("device", C.c_int32), ("window_bits", C.c_uint32), ("flags", C.c_uint32),
                ("stream_chunk_log2", C.c_uint32), ("max_points", C.c_uint64), ("batch_layout", C.c_uint32), ("host_threads", C.c_uint32)]


class Plan(C.Structure):
    _fields_ = [("window_bits", C.c_uint32), ("num_windows", C.c_uint32), ("num_buckets", C.c_uint32),
                ("signed_digits", C.c_uint32), ("workspace_bytes", C.c_uint64), ("virtual_points", C.c_uint64),
                ("glv", C.c_uint32), ("scalar_bits", C.c_uint32), ("table_factor", C.c_uint32), ("bucket_arrays", C.c_uint32),
                ("table_bytes", C.c_uint64), ("top_digit_bits", C.c_uint32), ("reserved", C.c_uint32)]


class Timings(C.Structure):
    _fields_ = [("h2d_ms", C.c_float), ("convert_ms", C.c_float), ("decompose_ms", C.c_float),
                ("sort_ms", C.c_float), ("accumulate_ms", C.c_float), ("reduce_ms", C.c_float),
                ("finish_ms", C.c_float), ("total_ms", C.c_float), ("num_points", C.c_uint64),
                ("num_adds", C.c_uint64), ("stream_chunks", C.c_uint32), ("batch_layout", C.c_uint32),
                ("plan_ms", C.c_float), ("combine_ms", C.c_float)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


_u32p = C.POINTER(C.c_uint32)
_u8p = C.POINTER(C.c_uint8)
_lib = None


def _preload_torch():
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7 / libhsa-runtime64 and the
    # library links the same soname from /opt/rocm.  Whichever is loaded first serves both, and loading
    # ours first leaves torch with a mixed runtime ("No HIP GPUs are available").  Python callers use torch
    # for device memory and torch.distributed, so let torch bring in its runtime before we dlopen.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass


def bind_product_abi(L):
    """ctypes signatures of every include/msm_hip.h entry point on a loaded library (product or hooks build)"""
    vp = C.c_void_p
    L.msm_abi_version.restype = C.c_uint32
    L.msm_ctx_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    L.msm_ctx_destroy.argtypes = [vp]
    L.msm_ctx_destroy.restype = None
    L.msm_last_error.argtypes = [vp]
    L.msm_last_error.restype = C.c_char_p
    L.msm_bn254_g1.argtypes = [vp, _u32p, C.c_uint32, _u8p, _u32p, C.c_size_t, _u32p, _u32p, _u8p]
    L.msm_bn254_g1_arkworks.argtypes = [vp, vp, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, _u32p, C.c_size_t, _u32p, _u32p, _u8p]
    L.msm_bn254_g1_upload_bases.argtypes = [vp, _u32p, C.c_uint32, _u8p, C.c_size_t]
    L.msm_bn254_g1_resident.argtypes = [vp, _u32p, C.c_size_t, _u32p, _u32p, _u8p]
    L.msm_bn254_g1_resident_batch.argtypes = [vp, C.POINTER(_u32p), C.c_size_t, C.c_size_t, _u32p, _u32p, _u8p]
    L.msm_tune_batch.argtypes = [vp, C.POINTER(_u32p), C.c_size_t, C.c_size_t, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_double)]
    L.msm_bn254_g1_device.argtypes = [vp, vp, vp, vp, C.c_size_t, vp, _u32p, _u32p, _u8p]
    L.msm_bn254_g1_resident_device.argtypes = [vp, vp, C.c_size_t, vp, _u32p, _u32p, _u8p]
    L.msm_bn254_g1_combine.argtypes = [_u32p, C.c_size_t, _u32p, _u32p, _u8p]
    L.msm_bn254_g1_combine_flags.argtypes = [_u32p, C.c_size_t, C.c_uint32, _u32p, _u32p, _u8p]
    L.msm_get_timings_sized.argtypes = [vp, vp, C.c_size_t]
    L.msm_multi_get_timings_sized.argtypes = [vp, C.c_int32, vp, C.c_size_t]
    L.msm_multi_set_kernel_timing.argtypes = [vp, C.c_uint32]
    L.msm_multi_get_clock_stats.argtypes = [vp, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    L.msm_plan.argtypes = [C.c_size_t, C.c_uint32, C.c_uint32, C.POINTER(Plan)]
    L.msm_get_timings.argtypes = [vp, C.POINTER(Timings)]
    L.msm_set_stage_timing.argtypes = [vp, C.c_int32]
    L.msm_set_kernel_timing.argtypes = [vp, C.c_uint32]
    L.msm_get_accumulate_kernel_stats.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    L.msm_reset_kernel_stats.argtypes = [vp]
    L.msm_reset_kernel_stats.restype = None
    L.msm_get_clock_stats.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    L.msm_bn254_g1_decompress.argtypes = [vp, _u8p, C.c_size_t, _u32p, _u8p, C.POINTER(C.c_int64)]
    L.msm_bn254_g1_upload_compressed.argtypes = [vp, _u8p, C.c_size_t, C.POINTER(C.c_int64)]
    L.msm_bn254_g1_compress.argtypes = [_u32p, C.c_uint32, _u8p, C.c_size_t, _u8p]
    L.msm_multi_create.argtypes = [C.POINTER(C.c_int32), C.c_int32, C.POINTER(Config), C.c_uint32, C.POINTER(vp)]
    L.msm_multi_destroy.argtypes = [vp]
    L.msm_multi_destroy.restype = None
    L.msm_multi_last_error.argtypes = [vp]
    L.msm_multi_last_error.restype = C.c_char_p
    L.msm_multi_num_devices.argtypes = [vp]
    L.msm_multi_exchange.argtypes = [vp]
    L.msm_multi_exchange.restype = C.c_uint32
    L.msm_bn254_g1_multi.argtypes = [vp, _u32p, C.c_uint32, _u8p, _u32p, C.c_size_t, _u32p, _u32p, _u8p]
    L.msm_bn254_g1_multi_arkworks.argtypes = [vp, vp, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, _u32p, C.c_size_t, _u32p, _u32p, _u8p]
    L.msm_bn254_g1_multi_device.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_size_t), _u32p, _u32p, _u8p]
    L.msm_multi_get_timings.argtypes = [vp, C.c_int32, C.POINTER(Timings)]
    L.msm_multi_get_exchange_stats.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int32]
    L.msm_multi_get_exchange_probe.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    for name in ABI_SYMBOLS:
        f = getattr(L, name)
        if f.restype is C.c_int:  # default
            f.restype = C.c_int32
    # the structs above (Config, Plan, Timings) are those of ONE ABI: a library of another would be read / written past their ends
    have = int(L.msm_abi_version())
    if have != ABI_VERSION:
        raise MsmError(ERR_STATE, "library ABI %d, this binding is written against ABI %d (rebuild: make -C gpu-acceleration_amd/csrc)" % (have, ABI_VERSION))
    return L


def load_library():
    """dlopen the in-tree libmsm_hip.so (the PRODUCT); fails loudly if it was not built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MsmError(ERR_NO_DEVICE, f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                      "(make -C gpu-acceleration_amd/csrc); there is no CPU fallback")
    _preload_torch()
    _lib = bind_product_abi(C.CDLL(LIB_PATH))
    return _lib


def _words(a, width):
    a = np.ascontiguousarray(a, dtype=np.uint32)
    return a.reshape(-1, width)


def _p32(a):
    return a.ctypes.data_as(_u32p) if a is not None else None


class MsmResult:
    """One G1 result: Jacobian Montgomery words (what the Rust shim turns into G1Projective via
    Fq::new_unchecked, metal_msm.rs:228-241) and the canonical affine standard-form words."""

    def __init__(self, jac, aff, inf):
        self.jacobian_mont = jac
        self._aff = aff
        self.is_infinity = bool(inf)

    @property
    def affine_std(self):
        """canonical affine standard-form words; computed on first use (one field inversion on the host) when the call
        returned only the Jacobian point -- the reference's own result type (metal_msm.rs:228-241)"""
        if self._aff is None:
            self._aff = combine_partials(self.jacobian_mont.reshape(1, 24))._aff
        return self._aff

    def affine_ints(self):
        if self.is_infinity:
            return None
        to_int = lambda ws: sum(int(w) << (32 * i) for i, w in enumerate(ws.tolist()))
        return to_int(self.affine_std[:8]), to_int(self.affine_std[8:])


def plan(n, window_bits=0, flags=0, _lib=None):
    p = Plan()
    rc = (_lib or load_library()).msm_plan(n, window_bits, flags, C.byref(p))
    if rc != OK:
        raise MsmError(rc, "Empty input" if rc == ERR_EMPTY else f"msm_plan failed ({rc})")
    return p


def combine_partials(partials_jacobian_mont, want_affine=True, flags=0):
    """Fold per-rank partial sums in fixed rank order (host arithmetic inside the library).
    want_affine=False skips the field inversion; MsmResult.affine_std then computes it on first use.
    flags=FLAG_DETERMINISTIC: the Jacobian words are the canonical Z = 1 representative (msm_bn254_g1_combine_flags) -- what the ranks of a
    one-process-per-GPU job fold with when their contexts carry the flag."""
    p = _words(partials_jacobian_mont, 24)
    jac, aff, inf = np.zeros(24, np.uint32), (np.zeros(16, np.uint32) if want_affine else None), C.c_uint8(0)
    rc = load_library().msm_bn254_g1_combine_flags(_p32(p), p.shape[0], flags, _p32(jac), _p32(aff), C.byref(inf))
    if rc != OK:
        raise MsmError(rc, "Empty input" if rc == ERR_EMPTY else f"combine failed ({rc})")
    return MsmResult(jac, aff, inf.value)


def compress_points(bases, form=FORM_STD, inf=None):
    """Host-side inverse of MsmContext.decompress: n x 16 coordinate words -> n x 32 bytes (no GPU needed)."""
    bases = _words(bases, 16)
    if bases.shape[0] == 0:
        raise MsmError(ERR_EMPTY, "Empty input")
    out = np.zeros(bases.shape[0] * 32, np.uint8)
    infp = None
    if inf is not None:
        inf = np.ascontiguousarray(inf, dtype=np.uint8)
        infp = inf.ctypes.data_as(_u8p)
    rc = load_library().msm_bn254_g1_compress(_p32(bases), form, infp, bases.shape[0], out.ctypes.data_as(_u8p))
    if rc != 0:
        raise MsmError(rc, "msm_bn254_g1_compress failed (%d)" % rc)
    return out.tobytes()


class MsmContext:
    """Persistent engine context (replaces MetalMSMPipeline, rebuilt per call in the reference)."""

    _loader = staticmethod(lambda: load_library())  # testhooks.HooksContext runs the same class on the hooks build

    def __init__(self, device=-1, window_bits=0, flags=0, max_points=0, stream_chunk_log2=0, batch_layout=BATCH_LAYOUT_AUTO, host_threads=0):
        self._lib = self._loader()
        cfg = Config(device, window_bits, flags, stream_chunk_log2, max_points, batch_layout, host_threads)
        h = C.c_void_p()
        rc = self._lib.msm_ctx_create(C.byref(cfg), C.byref(h))
        if rc != OK:
            raise MsmError(rc, (self._lib.msm_last_error(None) or b"").decode())
        self._h = h
        self.window_bits, self.flags = window_bits, flags
        # the device-pointer calls (the bench's timed call) write into one persistent buffer whose ctypes pointers are made once: building two numpy
        # arrays and their ctypes views cost 3 us per call, a copy of 96 bytes 0.2 (the C side serialises a context's calls; this lock covers the copy)
        import threading
        self._jbuf = np.zeros(24, np.uint32)
        self._jptr, self._oi = _p32(self._jbuf), C.c_uint8(0)
        self._oiref, self._olock = C.byref(self._oi), threading.Lock()

    def close(self):
        if getattr(self, "_h", None):
            self._lib.msm_ctx_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc):
        if rc != OK:
            raise MsmError(rc, (self._lib.msm_last_error(self._h) or b"").decode() or f"status {rc}")

    def _outs(self):
        return np.zeros(24, np.uint32), np.zeros(16, np.uint32), C.c_uint8(0)

    # -- the drop-in call ---------------------------------------------------------------------
    def msm(self, bases, scalars, form=FORM_STD, inf=None):
        bases, scalars = _words(bases, 16), _words(scalars, 8)
        if bases.shape[0] == 0 or scalars.shape[0] == 0:
            raise MsmError(ERR_EMPTY, "Empty input")  # metal_msm.rs:647-649
        n = min(bases.shape[0], scalars.shape[0])  # metal_msm.rs:652-656
        infp = None
        if inf is not None:
            inf = np.ascontiguousarray(inf, dtype=np.uint8)
            infp = inf.ctypes.data_as(_u8p)
        jac, aff, oi = self._outs()
        self._check(self._lib.msm_bn254_g1(self._h, _p32(bases), form, infp, _p32(scalars), n, _p32(jac), _p32(aff),
                                           C.byref(oi)))
        return MsmResult(jac, aff, oi.value)

    def msm_arkworks(self, raw_structs, stride, x_off, y_off, inf_off, scalars_mont):
        """Zero-copy path of the Rust shim: `raw_structs` is the byte image of a [G1Affine] slice, scalars are Fr
        Montgomery words; both go to the GPU untouched."""
        raw = np.ascontiguousarray(raw_structs, dtype=np.uint8).reshape(-1)
        sc = _words(scalars_mont, 8)
        n = min(raw.size // stride, sc.shape[0])
        if n == 0:
            raise MsmError(ERR_EMPTY, "Empty input")
        jac, aff, oi = self._outs()
        self._check(self._lib.msm_bn254_g1_arkworks(self._h, raw.ctypes.data_as(C.c_void_p), stride, x_off, y_off,
                                                    inf_off if inf_off is not None else C.c_size_t(-1).value, _p32(sc), n,
                                                    _p32(jac), _p32(aff), C.byref(oi)))
        return MsmResult(jac, aff, oi.value)

    def upload_bases(self, bases, form=FORM_STD, inf=None):
        bases = _words(bases, 16)
        if bases.shape[0] == 0:
            raise MsmError(ERR_EMPTY, "Empty input")
        infp = None
        if inf is not None:
            inf = np.ascontiguousarray(inf, dtype=np.uint8)
            infp = inf.ctypes.data_as(_u8p)
        self._check(self._lib.msm_bn254_g1_upload_bases(self._h, _p32(bases), form, infp, bases.shape[0]))

    @staticmethod
    def _images(images):
        buf = np.frombuffer(images, dtype=np.uint8) if isinstance(images, (bytes, bytearray, memoryview)) else \
            np.ascontiguousarray(images, dtype=np.uint8).reshape(-1)
        if buf.size == 0:
            raise MsmError(ERR_EMPTY, "Empty input")
        if buf.size % 32:
            raise MsmError(ERR_BAD_ARG, "compressed G1Affine images are 32 bytes each")
        return np.ascontiguousarray(buf)

    def decompress(self, images):
        """arkworks-0.4 `serialize_compressed` G1Affine images (n x 32 bytes) -> (xy Montgomery words n x 16, inf n).
        The square roots run on the GPU.  An image that does not decode raises MsmError(ERR_INVALID_DATA) with
        .first_invalid set (arkworks: SerializationError::InvalidData)."""
        buf = self._images(images)
        n = buf.size // 32
        xy = np.zeros((n, 16), np.uint32)
        inf = np.zeros(n, np.uint8)
        bad = C.c_int64(-1)
        rc = self._lib.msm_bn254_g1_decompress(self._h, buf.ctypes.data_as(_u8p), n, _p32(xy), inf.ctypes.data_as(_u8p), C.byref(bad))
        self._check_invalid(rc, bad)
        return xy, inf

    def upload_compressed(self, images):
        """Decode compressed images straight into the resident-bases set (then msm_resident)."""
        buf = self._images(images)
        bad = C.c_int64(-1)
        rc = self._lib.msm_bn254_g1_upload_compressed(self._h, buf.ctypes.data_as(_u8p), buf.size // 32, C.byref(bad))
        self._check_invalid(rc, bad)

    def _check_invalid(self, rc, bad):
        if rc == ERR_INVALID_DATA:
            e = MsmError(rc, (self._lib.msm_last_error(self._h) or b"invalid data").decode())
            e.first_invalid = int(bad.value)
            raise e
        self._check(rc)

    def msm_resident(self, scalars):
        scalars = _words(scalars, 8)
        if scalars.shape[0] == 0:
            raise MsmError(ERR_EMPTY, "Empty input")
        jac, aff, oi = self._outs()
        self._check(self._lib.msm_bn254_g1_resident(self._h, _p32(scalars), scalars.shape[0], _p32(jac), _p32(aff),
                                                    C.byref(oi)))
        return MsmResult(jac, aff, oi.value)

    def msm_resident_device(self, d_scalars_ptr, n, stream=None):
        """scalars already in HBM (raw device pointer) against the resident bases and, if the context has one, their window table"""
        if n == 0:
            raise MsmError(ERR_EMPTY, "Empty input")
        jac, _, oi = self._outs()
        self._check(self._lib.msm_bn254_g1_resident_device(self._h, d_scalars_ptr, n, stream, _p32(jac), None, C.byref(oi)))
        return MsmResult(jac, None, oi.value)

    def msm_resident_batch(self, scalar_vectors, want_affine=True):
        """several scalar vectors against the resident bases, two MSMs in flight (how provers call MSM): list of MsmResult"""
        vecs = [_words(s, 8) for s in scalar_vectors]
        if not vecs or any(v.shape[0] == 0 for v in vecs):
            raise MsmError(ERR_EMPTY, "Empty input")
        n = min(v.shape[0] for v in vecs)
        k = len(vecs)
        ptrs = (_u32p * k)(*[_p32(v) for v in vecs])
        jac = np.zeros((k, 24), np.uint32)
        aff = np.zeros((k, 16), np.uint32) if want_affine else None
        inf = np.zeros(k, np.uint8)
        self._check(self._lib.msm_bn254_g1_resident_batch(self._h, ptrs, n, k, _p32(jac), _p32(aff), inf.ctypes.data_as(_u8p)))
        return [MsmResult(jac[i], aff[i] if want_affine else None, inf[i]) for i in range(k)]

    def tune_batch(self, scalar_vectors, reps=0):
        """explicit, opt-in measurement of the batch layout on this context as the process is now (msm_tune_batch): returns
        (chosen MSM_BATCH_LAYOUT_*, {layout: ms per MSM}); AUTO contexts use the choice until the next upload"""
        vecs = [_words(s, 8) for s in scalar_vectors]
        if len(vecs) < 2 or any(v.shape[0] == 0 for v in vecs):
            raise MsmError(ERR_BAD_ARG, "tune_batch needs at least two non-empty scalar vectors")
        n = min(v.shape[0] for v in vecs)
        k = len(vecs)
        ptrs = (_u32p * k)(*[_p32(v) for v in vecs])
        chosen, ms = C.c_uint32(0), (C.c_double * 3)()
        self._check(self._lib.msm_tune_batch(self._h, ptrs, n, k, reps, C.byref(chosen), ms))
        return int(chosen.value), {BATCH_LAYOUT_ONE_STREAM: ms[0], BATCH_LAYOUT_ONE_STREAM_REDUCE: ms[1], BATCH_LAYOUT_TWO_STREAMS: ms[2]}

    def msm_device(self, d_bases_ptr, d_scalars_ptr, n, d_inf_ptr=None, stream=None):
        """All operands already in HBM (raw device pointers, e.g. torch.Tensor.data_ptr())."""
        if n == 0:
            raise MsmError(ERR_EMPTY, "Empty input")
        with self._olock:
            rc = self._lib.msm_bn254_g1_device(self._h, d_bases_ptr, d_inf_ptr, d_scalars_ptr, n, stream, self._jptr, None, self._oiref)
            if rc != OK:
                self._check(rc)
            return MsmResult(self._jbuf.copy(), None, self._oi.value)  # affine words on demand (MsmResult.affine_std)

    def set_stage_timing(self, enabled=True):
        self._check(self._lib.msm_set_stage_timing(self._h, int(bool(enabled))))

    def timings(self):
        t = Timings()
        self._check(self._lib.msm_get_timings(self._h, C.byref(t)))
        return t.as_dict()

    def set_kernel_timing(self, every_n):
        """every `every_n`-th launch of the accumulate kernel carries its pair of hipEvents (0 = none: the default; 1 = all: ~11 us more per MSM); accumulate_kernel_stats() and timings()["accumulate_ms"] see the timed launches only"""
        self._check(self._lib.msm_set_kernel_timing(self._h, every_n))

    def accumulate_kernel_stats(self):
        avg, cnt = C.c_double(0), C.c_uint64(0)
        self._check(self._lib.msm_get_accumulate_kernel_stats(self._h, C.byref(avg), C.byref(cnt)))
        return avg.value, cnt.value

    def reset_kernel_stats(self):
        self._lib.msm_reset_kernel_stats(self._h)

    def clock_stats(self):
        """clock probe of k_accumulate since the last reset_kernel_stats: {"sclk_ghz": shader clock the kernel sustained,
        "cycles_per_addition": shader cycles of one wavefront per mixed addition, "samples": launches sampled}"""
        ghz, cpa, cnt = C.c_double(0), C.c_double(0), C.c_uint64(0)
        self._check(self._lib.msm_get_clock_stats(self._h, C.byref(ghz), C.byref(cpa), C.byref(cnt)))
        return {"sclk_ghz": ghz.value, "cycles_per_addition": cpa.value, "samples": cnt.value}


class MsmMulti:
    """One MSM over several GPUs of ONE process (include/msm_hip.h "multi-GPU"): contiguous point-range shards, one context
    and one host thread per device, partials exchanged with RCCL (all-gather of 24 words + fold in rank order) or folded on
    the host.  Same call signatures as MsmContext."""

    def __init__(self, devices=None, window_bits=0, flags=0, stream_chunk_log2=0, exchange=EXCHANGE_AUTO, host_threads=0, _lib=None):
        self._lib = _lib or load_library()
        cfg = Config(-1, window_bits, flags, stream_chunk_log2, 0, 0, host_threads)
        h = C.c_void_p()
        if devices is None:
            rc = self._lib.msm_multi_create(None, 0, C.byref(cfg), exchange, C.byref(h))
        else:
            arr = (C.c_int32 * len(devices))(*devices)
            rc = self._lib.msm_multi_create(arr, len(devices), C.byref(cfg), exchange, C.byref(h))
        if rc != OK:
            raise MsmError(rc, (self._lib.msm_multi_last_error(None) or b"").decode())
        self._h = h
        self.num_devices = int(self._lib.msm_multi_num_devices(h))
        self.exchange = int(self._lib.msm_multi_exchange(h))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.msm_multi_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc):
        if rc != OK:
            raise MsmError(rc, (self._lib.msm_multi_last_error(self._h) or b"").decode() or f"status {rc}")

    def msm(self, bases, scalars, form=FORM_STD, inf=None):
        bases, scalars = _words(bases, 16), _words(scalars, 8)
        if bases.shape[0] == 0 or scalars.shape[0] == 0:
            raise MsmError(ERR_EMPTY, "Empty input")
        n = min(bases.shape[0], scalars.shape[0])
        infp = None
        if inf is not None:
            inf = np.ascontiguousarray(inf, dtype=np.uint8)
            infp = inf.ctypes.data_as(_u8p)
        jac, aff, oi = np.zeros(24, np.uint32), np.zeros(16, np.uint32), C.c_uint8(0)
        self._check(self._lib.msm_bn254_g1_multi(self._h, _p32(bases), form, infp, _p32(scalars), n, _p32(jac), _p32(aff), C.byref(oi)))
        return MsmResult(jac, aff, oi.value)

    def msm_arkworks(self, raw_structs, stride, x_off, y_off, inf_off, scalars_mont):
        raw = np.ascontiguousarray(raw_structs, dtype=np.uint8).reshape(-1)
        sc = _words(scalars_mont, 8)
        n = min(raw.size // stride, sc.shape[0])
        if n == 0:
            raise MsmError(ERR_EMPTY, "Empty input")
        jac, aff, oi = np.zeros(24, np.uint32), np.zeros(16, np.uint32), C.c_uint8(0)
        self._check(self._lib.msm_bn254_g1_multi_arkworks(self._h, raw.ctypes.data_as(C.c_void_p), stride, x_off, y_off,
                                                          inf_off if inf_off is not None else C.c_size_t(-1).value, _p32(sc), n,
                                                          _p32(jac), _p32(aff), C.byref(oi)))
        return MsmResult(jac, aff, oi.value)

    def msm_device(self, d_bases_ptrs, d_scalars_ptrs, counts, d_inf_ptrs=None):
        """shard g already sits in device g's HBM: raw device pointers and point counts per device"""
        G = self.num_devices
        assert len(d_bases_ptrs) == G and len(d_scalars_ptrs) == G and len(counts) == G
        vpa = C.c_void_p * G
        pb, ps = vpa(*d_bases_ptrs), vpa(*d_scalars_ptrs)
        pi = vpa(*d_inf_ptrs) if d_inf_ptrs is not None else None
        cn = (C.c_size_t * G)(*counts)
        jac, oi = np.zeros(24, np.uint32), C.c_uint8(0)
        self._check(self._lib.msm_bn254_g1_multi_device(self._h, pb, pi, ps, cn, _p32(jac), None, C.byref(oi)))
        return MsmResult(jac, None, oi.value)

    def timings(self, g=0):
        t = Timings()
        self._check(self._lib.msm_multi_get_timings(self._h, g, C.byref(t)))
        return t.as_dict()

    def exchange_stats(self):
        """(exchange ms of rank 0, [wall-clock ms of every rank's local MSM]) of the last call"""
        ex = C.c_float(0)
        sh = (C.c_float * self.num_devices)()
        self._check(self._lib.msm_multi_get_exchange_stats(self._h, C.byref(ex), sh, self.num_devices))
        return float(ex.value), [float(v) for v in sh]

    def set_kernel_timing(self, every_n):
        self._check(self._lib.msm_multi_set_kernel_timing(self._h, every_n))

    def clock_stats(self, g=0):
        """msm_get_clock_stats of rank g's context: {"sclk_ghz", "cycles_per_addition", "samples"}"""
        a, b, n = C.c_double(0), C.c_double(0), C.c_uint64(0)
        self._check(self._lib.msm_multi_get_clock_stats(self._h, g, C.byref(a), C.byref(b), C.byref(n)))
        return {"sclk_ghz": a.value, "cycles_per_addition": b.value, "samples": n.value}

    def exchange_probe(self):
        """(rccl ms, host-fold ms) per exchange that EXCHANGE_AUTO measured when the handle was created; (0, 0) = nothing was probed"""
        a, b = C.c_float(0), C.c_float(0)
        self._check(self._lib.msm_multi_get_exchange_probe(self._h, C.byref(a), C.byref(b)))
        return float(a.value), float(b.value)


_default_ctx = None


def default_context():
    """Process-global lazily created context, as the Rust shim keeps."""
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = MsmContext()
    return _default_ctx


def hip_variable_base_msm(bases, scalars, form=FORM_STD, inf=None):
    """Drop-in for metal_variable_base_msm(&bases, &scalars) (metal_msm.rs:642-695)."""
    return default_context().msm(bases, scalars, form, inf)


metal_variable_base_msm = hip_variable_base_msm  # the reference's own name, kept as an alias
