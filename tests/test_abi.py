"""CPU tests of the drop-in boundary: the C-ABI library loads without a GPU and exports every
symbol include/msm_hip.h declares; host-only entry points (planner, partial combine, input
streams) behave; compute entry points fail loudly -- never fall back -- when no GPU is present."""
import os
import re

import numpy as np
import pytest

import mopro_msm_hip as mh
from mopro_msm_hip import testhooks as th
from conftest import ROOT, load_golden
from oracle import bn254_oracle as orc


def header_symbols(name="msm_hip.h"):
    txt = open(os.path.join(ROOT, "include", name)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(msm_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    lib = mh.load_library()
    syms = header_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/msm_hip.h but not exported"
    assert sorted(mh.ABI_SYMBOLS) == syms
    assert lib.msm_abi_version() == mh.ABI_VERSION == 7


def test_hooks_live_in_their_own_library():
    """include/msm_hip_testhooks.h: exported by libmsm_hip_hooks.so (which also carries the whole product ABI), by the product
    library not at all"""
    prod, hooks = mh.load_library(), th.load_hooks_library()
    hs = header_symbols("msm_hip_testhooks.h")
    assert sorted(th.HOOK_SYMBOLS) == hs and len(hs) >= 7
    for s in hs:
        assert hasattr(hooks, s), f"{s} declared in include/msm_hip_testhooks.h but not exported by the hooks build"
        assert not hasattr(prod, s), f"{s} is a test hook and must not be exported by the product library"
    for s in header_symbols():
        assert hasattr(hooks, s)


def test_planner_matches_design():
    # up to 2^20 points: GLV split -- 2n virtual points, 127-bit halves, half the windows
    p = mh.plan(1 << 17)
    assert p.signed_digits == 1 and p.num_buckets == 1 << (p.window_bits - 1)
    assert p.glv == 1 and p.scalar_bits == 127 and p.virtual_points == 2 << 17
    assert p.num_windows == 127 // p.window_bits + 1
    assert 13 <= p.window_bits <= 17
    # without the split, and above 2^20 points (where it no longer pays): the reference's shape
    p = mh.plan(1 << 17, 0, mh.FLAG_NO_GLV)
    assert p.glv == 0 and p.scalar_bits == 254 and p.virtual_points == 1 << 17 and p.num_windows == 254 // p.window_bits + 1
    p = mh.plan(1 << 20)  # the BASELINE size: still split (round 3) -- eight 16-bit windows over 2^21 virtual points
    assert p.glv == 1 and p.window_bits == 16 and p.num_windows == 8 and p.virtual_points == 2 << 20
    p = mh.plan(1 << 20, 0, mh.FLAG_NO_GLV)
    assert p.glv == 0 and p.window_bits == 16 and p.num_windows == 16 and p.virtual_points == 1 << 20
    p = mh.plan(1 << 21)
    assert p.glv == 0 and p.window_bits == 17 and p.num_windows == 15
    assert mh.plan(1 << 20).glv == 1 and mh.plan((1 << 20) + 1).glv == 0
    # between 2^20 and 2^21 points: unsplit, still 16-bit windows (17 from 2^21 on)
    assert [mh.plan(n).window_bits for n in ((1 << 20) + 1, 1500000, (1 << 21) - 1, 1 << 21)] == [16, 16, 16, 17]
    # BASELINE config 2: fixed 16-bit window, plain digits
    p = mh.plan(1 << 16, 16, mh.FLAG_UNSIGNED_DIGITS | mh.FLAG_NO_GLV)
    assert (p.window_bits, p.num_windows, p.num_buckets, p.signed_digits) == (16, 16, 65536, 0)
    p = mh.plan(1 << 16, 16, mh.FLAG_UNSIGNED_DIGITS)
    assert (p.window_bits, p.num_windows, p.num_buckets, p.signed_digits, p.glv) == (16, 8, 65536, 0, 1)
    # row f4: MSM_FLAG_WINDOW_TABLE -- the plan of a resident call on a set uploaded under that flag (host-only: no GPU needed)
    T = mh.FLAG_WINDOW_TABLE
    p = mh.plan(1 << 20, 0, T)
    assert (p.window_bits, p.num_windows, p.table_factor, p.bucket_arrays, p.glv) == (20, 13, 13, 1, 0)
    assert p.num_buckets == 1 << 19 and p.table_bytes == 13 * (1 << 20) * 64 and p.workspace_bytes > p.table_bytes
    p = mh.plan(1 << 17, 0, T)  # up to 2^18 points: split, eight 16-bit windows in ONE array of 2^15 buckets
    assert (p.window_bits, p.num_windows, p.table_factor, p.bucket_arrays, p.glv, p.virtual_points) == (16, 8, 8, 1, 1, 2 << 17)
    assert mh.plan((1 << 18) + 1, 0, T).glv == 0 and mh.plan(1 << 21, 0, T).table_factor == 13
    assert mh.plan(1 << 22, 0, T).table_factor == 1                            # 13 x 2^22 entries: beyond TABLE_MAX_ENTRIES, no table
    assert mh.plan(1 << 16, 0, T | mh.FLAG_UNSIGNED_DIGITS).table_factor == 1   # plain digits: no table
    p, q = mh.plan(1 << 16, 0, 0), mh.plan(1 << 16, 0, T)
    assert p.table_factor == 1 and p.bucket_arrays == p.num_windows and p.table_bytes == 0 and q.num_windows == p.num_windows
    # reference table values are accepted as overrides (metal_msm.rs:661-673)
    for w, W in ((8, 32), (13, 20), (15, 17), (16, 16)):
        p = mh.plan(1 << 16, w, mh.FLAG_NO_GLV)
        assert p.num_windows == W and p.num_buckets == 1 << (w - 1)
    with pytest.raises(mh.MsmError) as e:
        mh.plan(0)
    assert e.value.code == mh.ERR_EMPTY
    with pytest.raises(mh.MsmError):
        mh.plan(16, 40)


def test_combine_partials_host_arithmetic():
    """msm_bn254_g1_combine == group sum of the partials (checked with the oracle)."""
    g = load_golden("rand_n17")
    _, _, j1 = orc.msm_pippenger(g["bases"][:9], g["scalars"][:9], orc.FORM_STD)
    _, _, j2 = orc.msm_pippenger(g["bases"][9:], g["scalars"][9:], orc.FORM_STD)
    r = mh.combine_partials(np.stack([j1, j2]))
    assert not r.is_infinity and (r.affine_std == g["expected"]).all()
    aff, inf = orc.g1_to_affine_std(r.jacobian_mont)
    assert inf == 0 and (aff == g["expected"]).all()
    # P + (-P) and identity partials
    neg = j1.copy()
    y = orc.words_to_int(j1[8:16])
    neg[8:16] = orc.int_to_words((orc.P - y) % orc.P)
    r = mh.combine_partials(np.stack([j1, neg]))
    assert r.is_infinity and not r.affine_std.any()
    ident = np.zeros(24, np.uint32)
    r = mh.combine_partials(np.stack([ident, j2, ident]))
    a2, _ = orc.g1_to_affine_std(j2)
    assert (r.affine_std == a2).all()
    r = mh.combine_partials(np.stack([j1, j1]))  # doubling branch
    a, _ = orc.g1_to_affine_std(orc.g1_dbl(j1))
    assert (r.affine_std == a).all()


def test_host_scalar_stream_matches_oracle_stream():
    a = th.generate_scalars_host(0xB2540002, 257)
    b = orc.gen_scalars(0xB2540002, 257)
    assert (a == b).all()
    a = th.generate_scalars_host(0xB2540001, 33, nonzero=True)
    assert (a == orc.gen_scalars(0xB2540001, 33, nonzero=True)).all()
    assert all(orc.words_to_int(x) < orc.R_ORDER for x in a)


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(mh.MsmError) as e:
        mh.MsmContext()
    assert e.value.code == mh.ERR_NO_DEVICE
    assert "no CPU fallback" in str(e.value)
    with pytest.raises(mh.MsmError):
        g = load_golden("rand_n3")
        mh.hip_variable_base_msm(g["bases"], g["scalars"])


def test_product_never_imports_oracle():
    """the product tree must not reference oracle/ (rule: oracle is test infrastructure)."""
    pkg = os.path.join(ROOT, "gpu-acceleration_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")) or f == "Makefile":
                txt = open(os.path.join(dp, f)).read()
                for line in txt.splitlines():
                    s = line.strip()
                    if s.startswith(("#include", "import", "from")) or "dlopen" in s or "CDLL" in s:
                        assert "oracle" not in s, (f, s)
