"""GPU tests (-m gpu) of the BASELINE.json configurations at their FULL per-GPU sizes and of the host-side paths around the
pipeline, by size-independent properties (closed form (sum s_i k_i mod r) * G on synthetic bases k_i * G):

  config 2  N = 2^16, fixed 16-bit window, plain digits, unsplit           test_config2_literal_shape
  config 4  2^24 over 8 GPUs = 2^21 points per GPU                          test_shard_sizes_of_configs_4_and_5[21]
  config 5  2^26 over 8 GPUs = 2^23 points per GPU, streamed                test_shard_sizes_of_configs_4_and_5[23],
                                                                            test_automatic_streaming_* (host->HBM chunks)
  N > 1     two ranks on ONE GPU through torch.distributed (gloo)           test_two_ranks_on_one_gpu
            two/three shards in ONE process through msm_multi              test_multi_*
  f2        msm_bn254_g1_resident_batch, two MSMs in flight                test_resident_batch_two_in_flight
  f1        msm_bn254_g1_arkworks at 2^20 with the (72, 0, 32, 64) layout  test_arkworks_entry_at_2_pow_20
Mirrors the reference's e2e test shape (tests/cuzk/e2e.rs:14-63: random instance, compare with the CPU result)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import mopro_msm_hip as mh
from mopro_msm_hip import testhooks as th
from conftest import ROOT, golden_cases, load_golden
from oracle import bn254_oracle as orc

pytestmark = pytest.mark.gpu
R = orc.R_ORDER


def _ints(words):
    a = np.ascontiguousarray(words, dtype=np.uint32).reshape(-1, 8)
    cols = [a[:, j].tolist() for j in range(8)]
    out = []
    for i in range(a.shape[0]):
        v = 0
        for j in range(7, -1, -1):
            v = (v << 32) | cols[j][i]
        out.append(v)
    return out


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _expected(dot):
    g = np.zeros(16, np.uint32)
    g[0], g[8] = 1, 2
    return orc.g1_to_affine_std(orc.g1_scalar_mul(g, orc.int_to_words(dot % R)))


class Instance:
    """synthetic instance in HBM (bases k_i*G as Montgomery words, scalars s_i) + the host-side logs for the closed form"""

    def __init__(self, hk, logn, seed=0xB2540031):
        import torch
        self.n = n = 1 << logn
        self.d_b = torch.empty(n * 16, dtype=torch.int32, device="cuda:0")
        self.d_s = torch.empty(n * 8, dtype=torch.int32, device="cuda:0")
        hk.generate_device(seed, seed + 1, n, self.d_b.data_ptr(), self.d_s.data_ptr())
        torch.cuda.synchronize()
        self.k = _ints(th.generate_scalars_host(seed, n, nonzero=True))
        self.s = _ints(th.generate_scalars_host(seed + 1, n))

    def dot(self, lo=0, hi=None, skip=None):
        hi = self.n if hi is None else hi
        if skip is None:
            return sum(a * b for a, b in zip(self.k[lo:hi], self.s[lo:hi]))
        return sum(a * b for i, (a, b) in enumerate(zip(self.k[lo:hi], self.s[lo:hi])) if not skip[lo + i])


@pytest.fixture(scope="module")
def hk():
    c = th.HooksContext()
    yield c
    c.close()


@pytest.fixture(scope="module")
def ctx():
    c = mh.MsmContext()
    yield c
    c.close()


@pytest.fixture(scope="module")
def inst20(hk):
    return Instance(hk, 20)


@pytest.mark.parametrize("logn", [21, 23])
def test_shard_sizes_of_configs_4_and_5(ctx, hk, logn):
    """the per-GPU shards of BASELINE config 4 (2^24 / 8) and config 5 (2^26 / 8), resident, closed form + a point-range split"""
    it = Instance(hk, logn, seed=0xB2540041 + logn)
    r = ctx.msm_device(it.d_b.data_ptr(), it.d_s.data_ptr(), it.n)
    exp, einf = _expected(it.dot())
    assert not r.is_infinity and einf == 0 and (r.affine_std == exp).all()
    assert mh.plan(it.n).glv == 0 and mh.plan(it.n).window_bits == 17  # 15 windows of 65536 buckets above 2^20 points
    h = it.n // 2 + 12345  # uneven split: the two halves of a 2-GPU run of twice the size
    p0 = ctx.msm_device(it.d_b.data_ptr(), it.d_s.data_ptr(), h)
    p1 = ctx.msm_device(it.d_b.data_ptr() + h * 64, it.d_s.data_ptr() + h * 32, it.n - h)
    assert (mh.combine_partials(np.stack([p0.jacobian_mont, p1.jacobian_mont])).affine_std == exp).all()


STREAM_MUL, MASK64 = 0xD1342543DE82EF95, (1 << 64) - 1  # element i of generator stream `seed` = SplitMix64 seeded with seed + i * STREAM_MUL


def _chunked_dot(seed, n, skip=None):
    """sum s_i k_i of the synthetic instance (hk.generate_device(seed, seed + 1, n, ...)), the host logs generated and multiplied in slices
    of 2^20 (numpy: orc.dot_words) -- a 2^24 instance never holds 2^25 Python integers"""
    dot = 0
    for c0 in range(0, n, 1 << 20):
        cnt = min(1 << 20, n - c0)
        k = th.generate_scalars_host((seed + c0 * STREAM_MUL) & MASK64, cnt, nonzero=True)
        sc = th.generate_scalars_host((seed + 1 + c0 * STREAM_MUL) & MASK64, cnt)
        if skip is not None:
            sc = sc.copy()
            sc[skip[c0:c0 + cnt] != 0] = 0
        dot += orc.dot_words(k, sc)
    return dot


def test_config4_full_instance_2_pow_24_resident_on_one_gpu(ctx, hk):
    """BASELINE config 4's WHOLE instance (2^24 points; the reference publishes 2^22 and 2^24 rows, README.md:86-93,
    106-114) inside the gate -- resident on one GPU (1.5 GB of inputs, ~20 ms of GPU time), cut by the engine into four point ranges of
    2^22 that accumulate INTO shared buckets, against the closed form."""
    import torch
    n, seed = 1 << 24, 0xB2540E01
    d_b = torch.empty(n * 16, dtype=torch.int32, device="cuda:0")
    d_s = torch.empty(n * 8, dtype=torch.int32, device="cuda:0")
    hk.generate_device(seed, seed + 1, n, d_b.data_ptr(), d_s.data_ptr())
    torch.cuda.synchronize()
    exp, einf = _expected(_chunked_dot(seed, n))
    for _ in range(2):
        r = ctx.msm_device(d_b.data_ptr(), d_s.data_ptr(), n)
        assert not r.is_infinity and einf == 0 and (r.affine_std == exp).all()
    tm = ctx.timings()
    assert tm["num_points"] == n and tm["stream_chunks"] == 4 and tm["num_adds"] > 14 * n
    del d_b, d_s
    torch.cuda.empty_cache()


def test_config5_shape_2_pow_22_host_streamed_arkworks_structs(ctx, hk):
    """2^22 points from HOST memory in arkworks' own layout -- 72-byte G1Affine structs (x, y, infinity) and Fr Montgomery
    words -- streamed host->HBM in chunks that overlap the accumulation (BASELINE config 5's mechanism at the size the reference
    publishes, README.md:91, 112), a few points at infinity, against the closed form."""
    import torch
    n, seed = 1 << 22, 0xB2540E11
    d_b = torch.empty(n * 16, dtype=torch.int32, device="cuda:0")
    d_s = torch.empty(n * 8, dtype=torch.int32, device="cuda:0")
    hk.generate_device(seed, seed + 1, n, d_b.data_ptr(), d_s.data_ptr())
    torch.cuda.synchronize()
    hb = d_b.cpu().numpy().view(np.uint32).reshape(n, 16)
    hs = d_s.cpu().numpy().view(np.uint32).reshape(n, 8)
    del d_b, d_s
    inf = np.zeros(n, np.uint8)
    inf[[0, 77, n // 2, n - 1]] = 1
    inf[(1 << 20) + 5:(1 << 20) + 9] = 1
    img = _ark_image(hb, inf=inf)
    # the scalar words are read as Fr MONTGOMERY form, i.e. as s_i * R^-1: the expected point follows by linearity
    exp, _ = _expected(_chunked_dot(seed, n, skip=inf) * pow(1 << 256, -1, R))
    r = ctx.msm_arkworks(img, 72, 0, 32, 64, hs)
    tm = ctx.timings()
    assert not r.is_infinity and (r.affine_std == exp).all()
    assert tm["stream_chunks"] >= 4 and tm["num_points"] == n


def test_config5_per_gpu_share_2_pow_23_streamed_from_host_memory(ctx, hk):
    """VERDICT r4 item 4 (what's missing 5): BASELINE config 5's share of ONE GPU at full size -- 2^23 points (2^26 / 8), 805 MB of arkworks
    words in pinned HOST memory, streamed host->HBM in 2^20-point chunks that overlap the accumulation INTO the shared buckets (c = 17,
    unsplit; the chunks' bases are gathered from the transfer slots as they arrived, round 5) -- against the closed form; a second call on
    PAGEABLE memory with an infinity mask gives the masked closed form."""
    import torch
    n, seed = 1 << 23, 0xB2540E21
    d_b = torch.empty(n * 16, dtype=torch.int32, device="cuda:0")
    d_s = torch.empty(n * 8, dtype=torch.int32, device="cuda:0")
    hk.generate_device(seed, seed + 1, n, d_b.data_ptr(), d_s.data_ptr())
    torch.cuda.synchronize()
    hb_t, hs_t = d_b.cpu().pin_memory(), d_s.cpu().pin_memory()
    del d_b, d_s
    torch.cuda.empty_cache()
    hb = hb_t.numpy().view(np.uint32).reshape(n, 16)
    hs = hs_t.numpy().view(np.uint32).reshape(n, 8)
    exp, einf = _expected(_chunked_dot(seed, n))
    r = ctx.msm(hb, hs, mh.FORM_MONT)
    tm = ctx.timings()
    assert not r.is_infinity and einf == 0 and (r.affine_std == exp).all()
    assert tm["stream_chunks"] == 8 and tm["num_points"] == n and tm["num_adds"] > 14 * n
    assert mh.plan(n).glv == 0 and mh.plan(n).window_bits == 17
    inf = np.zeros(n, np.uint8)
    inf[[0, 1, n // 3, n - 1]] = 1
    inf[(1 << 22) - 3:(1 << 22) + 3] = 1  # across a chunk border
    exp2, _ = _expected(_chunked_dot(seed, n, skip=inf))
    r2 = ctx.msm(np.array(hb), np.array(hs), mh.FORM_MONT, inf)  # pageable copies
    assert (r2.affine_std == exp2).all() and ctx.timings()["stream_chunks"] == 8
    del hb_t, hs_t


def test_config2_literal_shape(hk):
    """BASELINE config 2 as literally stated: N = 2^16, fixed 16-bit window, plain (unsigned) digits, no GLV split --
    W = 16 windows of 65536 buckets"""
    it = Instance(hk, 16, seed=0xB2540051)
    flags = mh.FLAG_UNSIGNED_DIGITS | mh.FLAG_NO_GLV
    p = mh.plan(it.n, 16, flags)
    assert (p.window_bits, p.num_windows, p.num_buckets, p.signed_digits, p.glv) == (16, 16, 65536, 0, 0)
    exp, _ = _expected(it.dot())
    with mh.MsmContext(window_bits=16, flags=flags) as c:
        for _ in range(2):
            r = c.msm_device(it.d_b.data_ptr(), it.d_s.data_ptr(), it.n)
            assert (r.affine_std == exp).all() and not r.is_infinity
    # and with the planner's own choice
    with mh.MsmContext() as c:
        assert (c.msm_device(it.d_b.data_ptr(), it.d_s.data_ptr(), it.n).affine_std == exp).all()


def test_automatic_streaming_pinned_and_pageable_at_2_pow_22(hk):
    """config 5's mechanism with the DEFAULT configuration (stream_chunk_log2 = 0): from 2^19 points on a host-pointer call is cut
    into chunks that travel on the copy stream while the previous chunk is accumulated INTO the shared buckets -- from pinned
    caller memory (asynchronous copies) and from pageable memory (the runtime stages them).  Same bits either way."""
    import torch
    it = Instance(hk, 22, seed=0xB2540061)
    exp, _ = _expected(it.dot())
    hb_t, hs_t = it.d_b.cpu(), it.d_s.cpu()
    hb = hb_t.numpy().view(np.uint32).reshape(it.n, 16)
    hs = hs_t.numpy().view(np.uint32).reshape(it.n, 8)
    hbp, hsp = hb_t.pin_memory(), hs_t.pin_memory()
    hbpn, hspn = hbp.numpy().view(np.uint32).reshape(it.n, 16), hsp.numpy().view(np.uint32).reshape(it.n, 8)
    with mh.MsmContext() as c:
        for b, s in ((hbpn, hspn), (hb, hs)):
            r = c.msm(b, s, mh.FORM_MONT)
            tm = c.timings()
            assert (r.affine_std == exp).all() and tm["stream_chunks"] >= 4 and tm["num_points"] == it.n
            assert tm["num_adds"] > 14 * it.n  # the running count covers every chunk (15 windows, ~1 - 2^-17 non-zero digits)
        # ragged sizes: a remainder below half a chunk joins the last chunk, a larger one is split
        for m in (it.n - 77777, it.n - (1 << 19) + 5, (1 << 21) + 3):
            e2, _ = _expected(it.dot(0, m))
            for b, s in ((hb, hs), (hbpn, hspn)):
                r = c.msm(b[:m], s[:m], mh.FORM_MONT)
                assert (r.affine_std == e2).all() and c.timings()["stream_chunks"] >= 4, m
        # an infinity mask travels with the chunks
        inf = np.zeros(it.n, np.uint8)
        inf[[3, 1 << 20, it.n - 1]] = 1
        e4, _ = _expected(it.dot(skip=inf))
        r = c.msm(hbpn, hspn, mh.FORM_MONT, inf)
        assert (r.affine_std == e4).all()
        # below the streaming threshold: single shot (bases travel on the copy stream beside the sort)
        m = 1 << 18
        e3, _ = _expected(it.dot(0, m))
        for b, s in ((hb, hs), (hbpn, hspn)):
            r = c.msm(b[:m], s[:m], mh.FORM_MONT)
            assert (r.affine_std == e3).all() and c.timings()["stream_chunks"] == 0


def test_pageable_caller_memory_is_pinned_in_place_for_the_call(hk, tmp_path):
    """round 6: a host-pointer call registers the caller's pageable arrays for its duration (hipHostRegister: the copies then run at the pinned rate) and
    unregisters them afterwards.  The registrations are reference-counted process-wide: TWO contexts handed the SAME arrays at the same time (two host threads)
    must both be right and leave the arrays usable; a read-only file mapping (a proving key mmap'ed from disk) must work whether or not it can be registered; the
    arrays are ordinary memory again afterwards (torch can pin a copy, numpy can write)."""
    import threading
    it = Instance(hk, 20, seed=0xB2540B01)
    exp, _ = _expected(it.dot())
    hb = it.d_b.cpu().numpy().view(np.uint32).reshape(it.n, 16).copy()
    hs = it.d_s.cpu().numpy().view(np.uint32).reshape(it.n, 8).copy()
    out, errs = {}, []

    def worker(tag):
        try:
            with mh.MsmContext() as c:
                for _ in range(4):
                    out.setdefault(tag, []).append(c.msm(hb, hs, mh.FORM_MONT).affine_std.copy())
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs
    assert all((r == exp).all() for k in out for r in out[k]) and sum(len(v) for v in out.values()) == 8
    hb[0, 0] ^= 0  # still writable, ordinary memory
    # a read-only mapping of the same bytes
    fb, fs = tmp_path / "bases.bin", tmp_path / "scalars.bin"
    hb.tofile(fb)
    hs.tofile(fs)
    mb = np.memmap(fb, dtype=np.uint32, mode="r").reshape(it.n, 16)
    ms = np.memmap(fs, dtype=np.uint32, mode="r").reshape(it.n, 8)
    with mh.MsmContext() as c:
        assert (c.msm(mb, ms, mh.FORM_MONT).affine_std == exp).all()
        assert (c.msm(mb[: 1 << 18], ms[: 1 << 18], mh.FORM_MONT).affine_std == _expected(it.dot(0, 1 << 18))[0]).all()
        c.upload_bases(mb, mh.FORM_MONT)
        assert (c.msm_resident(ms).affine_std == exp).all()
        assert all((r.affine_std == exp).all() for r in c.msm_resident_batch([ms, hs, ms], want_affine=True))
    with mh.MsmMulti(devices=[0, 0]) as m:  # the handle registers the arrays once for both ranks
        assert (m.msm(hb, hs, mh.FORM_MONT).affine_std == exp).all()
        assert (m.msm(mb, ms, mh.FORM_MONT).affine_std == exp).all()


def test_streamed_shared_buckets_small_chunks(hk):
    """forced tiny chunks: GLV plan of the whole instance shared by all chunks, infinity masks, a scalar error in a
    late chunk"""
    n = 1 << 17
    k = orc.gen_scalars(0xB2540071, n, nonzero=True)
    s = orc.gen_scalars(0xB2540072, n)
    bases = orc.gen_bases_from_logs(k, orc.FORM_MONT)
    inf = np.zeros(n, np.uint8)
    inf[[0, 5, 40000, n - 1]] = 1
    kk, ss = _ints(k), _ints(s)
    exp, _ = _expected(sum(a * b for i, (a, b) in enumerate(zip(kk, ss)) if not inf[i]))
    for lg in (12, 15):
        with mh.MsmContext(stream_chunk_log2=lg) as c:
            r = c.msm(bases, s, mh.FORM_MONT, inf)
            tm = c.timings()
            assert (r.affine_std == exp).all(), lg
            assert tm["stream_chunks"] == n >> lg
            assert mh.plan(n).glv == 1  # 2n virtual points per chunk, one bucket array
            bad = s.copy()
            bad[n - 3, 7] |= 0x40000000
            with pytest.raises(mh.MsmError) as e:
                c.msm(bases, bad, mh.FORM_MONT, inf)
            assert e.value.code == mh.ERR_BAD_ARG
            r = c.msm(bases, s, mh.FORM_MONT, inf)  # the context recovers
            assert (r.affine_std == exp).all()
    with mh.MsmContext(stream_chunk_log2=13, flags=mh.FLAG_NO_GLV, window_bits=13) as c:
        assert (c.msm(bases, s, mh.FORM_MONT, inf).affine_std == exp).all()


def _ark_image(hb, stride=72, x_off=0, y_off=32, inf_off=64, inf=None):
    n = hb.shape[0]
    img = np.zeros((n, stride), np.uint8)
    raw = hb.view(np.uint8).reshape(n, 64)
    img[:, x_off:x_off + 32] = raw[:, :32]
    img[:, y_off:y_off + 32] = raw[:, 32:]
    if inf is not None and inf_off is not None:
        img[:, inf_off] = inf
    return img


def test_arkworks_entry_at_2_pow_20(ctx, inst20):
    """msm_bn254_g1_arkworks at the BASELINE size with arkworks' real layout shape (72-byte G1Affine: x, y, infinity), a few
    infinity flags, Fr words in Montgomery form.  The scalar words are taken AS Montgomery words, i.e. the scalars are
    s_i * 2^-256 mod r: by linearity the expected point is (2^-256 * sum s_i k_i) * G.  Streamed (>= 2^19 points) and,
    on a 2^18 prefix, single shot."""
    it = inst20
    hb = it.d_b.cpu().numpy().view(np.uint32).reshape(it.n, 16)
    hs = it.d_s.cpu().numpy().view(np.uint32).reshape(it.n, 8)
    inf = np.zeros(it.n, np.uint8)
    inf[[1, 77, 300000, it.n - 1]] = 1
    img = _ark_image(hb, inf=inf)
    rinv = pow(1 << 256, -1, R)
    exp, _ = _expected(it.dot(skip=inf) * rinv)
    r = ctx.msm_arkworks(img, 72, 0, 32, 64, hs)
    assert (r.affine_std == exp).all() and ctx.timings()["stream_chunks"] >= 2
    m = 1 << 18
    e2, _ = _expected(it.dot(0, m, skip=inf) * rinv)
    r = ctx.msm_arkworks(img[:m], 72, 0, 32, 64, hs[:m])
    assert (r.affine_std == e2).all() and ctx.timings()["stream_chunks"] == 0


def test_resident_set_survives_other_calls(ctx, inst20):
    """upload_bases -> msm_device on OTHER bases -> msm_resident must still use the uploaded set (the resident bases
    live in their own buffers; device / host calls use scratch)"""
    it = inst20
    n = 1 << 16
    k = orc.gen_scalars(0xB2540081, n, nonzero=True)
    s = orc.gen_scalars(0xB2540082, n)
    bases = orc.gen_bases_from_logs(k, orc.FORM_MONT)
    exp, _ = orc.closed_form_expected(k, s)
    ctx.upload_bases(bases, mh.FORM_MONT)
    assert (ctx.msm_resident(s).affine_std == exp).all()
    ctx.msm_device(it.d_b.data_ptr(), it.d_s.data_ptr(), it.n)                      # other bases, larger
    ctx.msm_device(it.d_b.data_ptr() + 64 * 999, it.d_s.data_ptr(), n)              # other bases, same size (GLV scratch)
    g = load_golden("rand_n1024")
    ctx.msm(g["bases"], g["scalars"], mh.FORM_STD, g["inf"])                          # host call with an infinity mask
    assert (ctx.msm_resident(s).affine_std == exp).all()
    e2, _ = orc.closed_form_expected(k[:5000], s[:5000])
    assert (ctx.msm_resident(s[:5000]).affine_std == e2).all()


def test_resident_batch_two_in_flight(ctx, inst20):
    """msm_bn254_g1_resident_batch: several scalar vectors against the resident bases with two MSMs in flight (second pipeline
    inside the context).  Every result equals the single-call result and the closed form; sizes on both sides of the GLV bound,
    a truncated batch (shorter scalar vectors), a batch of one, edge scalar vectors (zeros, all r-1), and single calls after."""
    for logn, count in ((16, 7), (20, 4)):
        n = 1 << logn
        if logn == 16:
            k = orc.gen_scalars(0xB2540091, n, nonzero=True)
            ctx.upload_bases(orc.gen_bases_from_logs(k, orc.FORM_MONT), mh.FORM_MONT)
        else:  # the 2^20 bases made on the GPU
            ctx.upload_bases(inst20.d_b.cpu().numpy().view(np.uint32).reshape(n, 16), mh.FORM_MONT)
        vecs = [th.generate_scalars_host(0xB25400A0 + 16 * logn + j, n) for j in range(count)]
        vecs[1] = np.zeros_like(vecs[1])
        vecs[2] = np.tile(orc.int_to_words(R - 1), (n, 1)).astype(np.uint32)
        res = ctx.msm_resident_batch(vecs)
        assert len(res) == count and res[1].is_infinity
        for j, (v, r) in enumerate(zip(vecs, res)):
            one = ctx.msm_resident(v)
            assert r.is_infinity == one.is_infinity and (r.affine_std == one.affine_std).all(), (logn, j)
            if logn == 16 and not r.is_infinity:
                exp, _ = orc.closed_form_expected(k, v)
                assert (r.affine_std == exp).all(), (logn, j)
        m = n - 4321  # shorter scalar vectors: truncation, the phi records are not used
        short = ctx.msm_resident_batch([v[:m] for v in vecs[:3]])
        for v, r in zip(vecs[:3], short):
            one = ctx.msm_resident(v[:m])
            assert r.is_infinity == one.is_infinity and (r.affine_std == one.affine_std).all()
        assert (ctx.msm_resident_batch([vecs[0]])[0].affine_std == res[0].affine_std).all()
    bad = vecs[3].copy()
    bad[777, 7] = 0xFFFFFFFF  # not a canonical Fr element: the batch stops with the error of that MSM, the context stays usable
    with pytest.raises(mh.MsmError) as e:
        ctx.msm_resident_batch([vecs[0], vecs[3], bad, vecs[0], vecs[3]])
    assert e.value.code == mh.ERR_BAD_ARG
    again = ctx.msm_resident_batch([vecs[0], vecs[3]])
    assert (again[0].affine_std == res[0].affine_std).all() and (again[1].affine_std == res[3].affine_std).all()
    with pytest.raises(mh.MsmError):
        ctx.msm_resident_batch([])
    c2 = mh.MsmContext()
    with pytest.raises(mh.MsmError) as e:
        c2.msm_resident_batch([vecs[0][:16]])
    assert e.value.code == mh.ERR_STATE
    c2.close()


def _batch_instance(hk, n, nvec):
    import torch
    dev = torch.device("cuda:0")
    d_bases = torch.empty(n * 16, dtype=torch.int32, device=dev)
    d_s = torch.empty(n * 8, dtype=torch.int32, device=dev)
    k = th.generate_scalars_host(0xB25400C1, n, nonzero=True)
    hk.generate_device(0xB25400C1, 0xB25400C2, n, d_bases.data_ptr(), d_s.data_ptr())
    torch.cuda.synchronize()
    hb = d_bases.cpu().numpy().view(np.uint32).reshape(n, 16)
    vecs = [th.generate_scalars_host(0xB25400D0 + j, n) for j in range(nvec)]
    exp = [orc.closed_form_expected(k, v)[0] for v in vecs]
    return hb, vecs, exp


@pytest.mark.parametrize("flags", [0, mh.FLAG_WINDOW_TABLE])
def test_resident_batch_layout_is_deterministic(hk, flags):
    """VERDICT r3 item 6 / ADVICE r3: the batch's stream layout is a pure function of (msm_config_t.batch_layout, the tuned choice, the
    clamped n) -- nothing is timed behind the caller's back.  Every forced layout gives the closed-form results; AUTO contexts created in
    either order, with other contexts alive, report the SAME layout (one stream from 2^19 points, two streams below); batches of
    alternating sizes and truncated calls keep reporting the layout of their own size class."""
    n = (1 << 19) + 4097
    hb, vecs, exp = _batch_instance(hk, n, 5)
    others = [mh.MsmContext() for _ in range(3)]
    try:
        for o in others:  # each with its second pipeline: the constellation that made round 3's measured default flip
            o.upload_bases(hb[:2048], mh.FORM_MONT)
            o.msm_resident_batch([vecs[0][:2048], vecs[1][:2048]])
        for layout in (mh.BATCH_LAYOUT_ONE_STREAM, mh.BATCH_LAYOUT_ONE_STREAM_REDUCE, mh.BATCH_LAYOUT_TWO_STREAMS):
            with mh.MsmContext(flags=flags, batch_layout=layout) as c:
                c.upload_bases(hb, mh.FORM_MONT)
                for off, cnt in ((0, 5), (3, 2), (1, 4)):
                    for j, r in enumerate(c.msm_resident_batch(vecs[off:off + cnt])):
                        assert (r.affine_std == exp[off + j]).all(), (flags, layout, off, cnt, j)
                    assert c.timings()["batch_layout"] == layout
        seen = []
        for order in (0, 1):
            a, b = mh.MsmContext(flags=flags), mh.MsmContext(flags=flags)
            first, second = (a, b) if order == 0 else (b, a)
            lay = []
            for c in (first, second):
                c.upload_bases(hb, mh.FORM_MONT)
                big = c.msm_resident_batch(vecs[:3])
                lay.append(c.timings()["batch_layout"])
                small = c.msm_resident_batch([v[:4096] for v in vecs[:2]])  # truncated call: n is clamped, the small class decides
                lay.append(c.timings()["batch_layout"])
                again = c.msm_resident_batch(vecs[1:4])                     # ... and the class switch does not disturb the big one
                lay.append(c.timings()["batch_layout"])
                assert all((r.affine_std == exp[j]).all() for j, r in enumerate(big))
                assert all((r.affine_std == exp[1 + j]).all() for j, r in enumerate(again))
                assert len(small) == 2
            seen.append(lay)
            a.close()
            b.close()
        want = [mh.BATCH_LAYOUT_ONE_STREAM, mh.BATCH_LAYOUT_TWO_STREAMS, mh.BATCH_LAYOUT_ONE_STREAM] * 2
        assert seen[0] == want and seen[1] == want, seen
    finally:
        for o in others:
            o.close()
    with pytest.raises(mh.MsmError) as e:
        mh.MsmContext(batch_layout=9)
    assert e.value.code == mh.ERR_BAD_ARG


def test_tune_batch_is_explicit_and_reset_by_an_upload(hk):
    """msm_tune_batch: the opt-in measurement.  It returns one of the three layouts and its three measurements, AUTO batch calls of that
    size class then run under it, the other size class and a context with a configured layout are not touched, and a new upload forgets it."""
    n = 1 << 17
    hb, vecs, exp = _batch_instance(hk, n, 4)
    layouts = (mh.BATCH_LAYOUT_ONE_STREAM, mh.BATCH_LAYOUT_ONE_STREAM_REDUCE, mh.BATCH_LAYOUT_TWO_STREAMS)
    with mh.MsmContext() as c:
        c.upload_bases(hb, mh.FORM_MONT)
        res = c.msm_resident_batch(vecs)
        assert c.timings()["batch_layout"] == mh.BATCH_LAYOUT_TWO_STREAMS  # AUTO below 2^19 points
        chosen, ms = c.tune_batch(vecs, reps=2)
        assert chosen in layouts and set(ms) == set(layouts) and all(v > 0 for v in ms.values())
        assert ms[chosen] == min(ms.values())
        res2 = c.msm_resident_batch(vecs)
        assert c.timings()["batch_layout"] == chosen
        for j in range(4):
            assert (res[j].affine_std == exp[j]).all() and (res2[j].affine_std == exp[j]).all()
        c.upload_bases(hb, mh.FORM_MONT)  # a new base set: the measurement belongs to the old one
        c.msm_resident_batch(vecs[:2])
        assert c.timings()["batch_layout"] == mh.BATCH_LAYOUT_TWO_STREAMS
        with pytest.raises(mh.MsmError):
            c.tune_batch(vecs[:1])
    with mh.MsmContext(batch_layout=mh.BATCH_LAYOUT_ONE_STREAM) as c:
        c.upload_bases(hb, mh.FORM_MONT)
        chosen, ms = c.tune_batch(vecs[:2], reps=1)
        assert chosen == mh.BATCH_LAYOUT_ONE_STREAM  # configured: kept, whatever was measured
        c.msm_resident_batch(vecs[:2])
        assert c.timings()["batch_layout"] == mh.BATCH_LAYOUT_ONE_STREAM


# ---- N > 1 -------------------------------------------------------------------------------------------------------------------
def test_multi_in_process_deterministic_flag_and_exchange_probe(inst20):
    """msm_multi with MSM_FLAG_DETERMINISTIC in its configuration: every per-device partial AND the fold are handed out as Z = 1 representatives
    -- ten calls on three ranks, host pointers and resident shards, give one set of 24 words.  On one physical device AUTO has nothing to measure:
    the probe reads (0, 0) and the exchange is the host fold; a single device with the exchange forced to RCCL is never probed either."""
    it = inst20
    exp, _ = _expected(it.dot())
    hb = it.d_b.cpu().numpy().view(np.uint32).reshape(it.n, 16)
    hs = it.d_s.cpu().numpy().view(np.uint32).reshape(it.n, 8)
    one = orc.fq_to_mont(orc.int_to_words(1))
    G = 3
    cuts = [g * it.n // G for g in range(G + 1)]
    with mh.MsmMulti(devices=[0] * G, flags=mh.FLAG_DETERMINISTIC) as m:
        assert m.exchange == mh.EXCHANGE_HOST and m.exchange_probe() == (0.0, 0.0)
        reps = set()
        for k in range(10):
            if k % 2:
                r = m.msm(hb, hs, mh.FORM_MONT)
            else:
                r = m.msm_device([it.d_b.data_ptr() + 64 * cuts[g] for g in range(G)], [it.d_s.data_ptr() + 32 * cuts[g] for g in range(G)],
                                 [cuts[g + 1] - cuts[g] for g in range(G)])
            assert (r.affine_std == exp).all() and (r.jacobian_mont[16:] == one).all()
            reps.add(r.jacobian_mont.tobytes())
        assert len(reps) == 1
    with mh.MsmMulti(devices=[0], exchange=mh.EXCHANGE_RCCL) as m:
        assert m.exchange == mh.EXCHANGE_RCCL and m.exchange_probe() == (0.0, 0.0)


def test_multi_in_process_on_one_gpu(hk, inst20):
    """msm_multi (include/msm_hip.h "multi-GPU"): one context + host thread per listed device.  On a 1-GPU box the list names
    device 0 several times, which selects the host fold of the partials; a single device with the exchange forced to RCCL runs
    the dlopen'ed ncclCommInitAll / ncclAllGather path with one rank."""
    it = inst20
    exp, _ = _expected(it.dot())
    hb = it.d_b.cpu().numpy().view(np.uint32).reshape(it.n, 16)
    hs = it.d_s.cpu().numpy().view(np.uint32).reshape(it.n, 8)
    for devices in ([0, 0], [0, 0, 0]):
        with mh.MsmMulti(devices=devices) as m:
            assert m.num_devices == len(devices) and m.exchange == mh.EXCHANGE_HOST
            for name in golden_cases():
                g = load_golden(name)
                r = m.msm(g["bases"], g["scalars"], mh.FORM_STD, g["inf"])  # includes n = 1, 2, 3 < ndev: idle ranks add the identity
                assert r.is_infinity == bool(g["expected_inf"]) and (r.affine_std == g["expected"]).all(), (name, devices)
            r = m.msm(hb, hs, mh.FORM_MONT)  # 2^20 host pointers: every shard streams its own range
            assert (r.affine_std == exp).all()
            assert sum(m.timings(g)["num_points"] for g in range(len(devices))) == it.n
            # shards already resident
            G = len(devices)
            cuts = [g * it.n // G for g in range(G + 1)]
            r = m.msm_device([it.d_b.data_ptr() + 64 * cuts[g] for g in range(G)], [it.d_s.data_ptr() + 32 * cuts[g] for g in range(G)],
                             [cuts[g + 1] - cuts[g] for g in range(G)])
            assert (r.affine_std == exp).all()
            # zero-copy arkworks structs, sharded
            img = _ark_image(hb[: 1 << 17])
            e2, _ = _expected(it.dot(0, 1 << 17) * pow(1 << 256, -1, R))
            assert (m.msm_arkworks(img, 72, 0, 32, 64, hs[: 1 << 17]).affine_std == e2).all()
            with pytest.raises(mh.MsmError) as e:
                m.msm(np.zeros((0, 16), np.uint32), np.zeros((0, 8), np.uint32))
            assert e.value.code == mh.ERR_EMPTY
    with pytest.raises(mh.MsmError) as e:
        mh.MsmMulti(devices=[0, 0], exchange=mh.EXCHANGE_RCCL)  # RCCL needs distinct devices
    assert e.value.code == mh.ERR_RCCL
    with pytest.raises(mh.MsmError):
        mh.MsmMulti(devices=[99])
    try:
        m = mh.MsmMulti(devices=[0], exchange=mh.EXCHANGE_RCCL)
    except mh.MsmError as err:  # no librccl on this box: the AUTO mode would have used the host fold
        assert err.code == mh.ERR_RCCL
    else:
        with m:
            assert m.exchange == mh.EXCHANGE_RCCL
            g = load_golden("rand_n1024")
            assert (m.msm(g["bases"], g["scalars"], mh.FORM_STD, g["inf"]).affine_std == g["expected"]).all()
            r = m.msm_device([it.d_b.data_ptr()], [it.d_s.data_ptr()], [it.n])
            assert (r.affine_std == exp).all()


def _torchrun_two_ranks(extra_env=None):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", **(extra_env or {}))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_gpu_worker.py")]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    return p, [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]


def test_two_ranks_on_one_gpu():
    """the torch.distributed path of bench.py / distributed.py with two REAL processes sharing cuda:0 (exchange over gloo):
    every rank must end with the same, correct bits (tests/dist_gpu_worker.py asserts on every rank)"""
    p, lines = _torchrun_two_ranks()
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert len(lines) == 2 and all(l["ok"] for l in lines) and lines[0]["affine"] == lines[1]["affine"]


def test_two_ranks_deterministic_flag_gives_the_canonical_jacobian_on_every_rank():
    """with MSM_FLAG_DETERMINISTIC on every rank's context the one-process-per-GPU fold (distributed.all_reduce_msm ->
    msm_bn254_g1_combine_flags) must return the Z = 1 representative -- the same 24 words on both ranks, on every repetition, and the
    words a single context with the flag returns for the whole instance."""
    p, lines = _torchrun_two_ranks({"MSM_TEST_DETERMINISTIC": "1"})
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert len(lines) == 2 and all(l["ok"] for l in lines), lines
    assert lines[0]["jacobian"] == lines[1]["jacobian"] and all(l["jacobian_representations"] == 1 for l in lines)
    # one context, whole instance (the worker's generator streams: seeds 0xB2540091 / 92, 2^18 points)
    import torch
    n = 1 << 18
    d_b = torch.empty(n * 16, dtype=torch.int32, device="cuda:0")
    d_s = torch.empty(n * 8, dtype=torch.int32, device="cuda:0")
    with th.HooksContext() as gen:
        gen.generate_device(0xB2540091, 0xB2540092, n, d_b.data_ptr(), d_s.data_ptr())
    with mh.MsmContext(flags=mh.FLAG_DETERMINISTIC) as c:
        r = c.msm_device(d_b.data_ptr(), d_s.data_ptr(), n)
    assert r.jacobian_mont.tolist() == lines[0]["jacobian"]
    # and the plain fold keeps handing out whatever representative the addition chain gives (flags = 0), the flagged one the canonical words
    parts = np.stack([r.jacobian_mont, np.zeros(24, np.uint32)])
    assert (mh.combine_partials(parts, flags=mh.FLAG_DETERMINISTIC).jacobian_mont == r.jacobian_mont).all()
    with pytest.raises(mh.MsmError):
        mh.combine_partials(parts, flags=mh.FLAG_WINDOW_TABLE)  # only the representative may be chosen here


@pytest.mark.parametrize("threads", [1, 2, 4])
def test_host_threads_of_the_configuration(hk, threads):
    """msm_config_t.host_threads (ABI 6, was `reserved`): 1 = no pool, the caller finishes alone; 2 = default; 4; 65 -> BAD_ARG.
    Same bits whatever the thread count, through a context and through msm_multi."""
    it = Instance(hk, 16, seed=0xB2540A51)
    exp, _ = _expected(it.dot())
    with mh.MsmContext(host_threads=threads) as c:
        for _ in range(3):
            assert (c.msm_device(it.d_b.data_ptr(), it.d_s.data_ptr(), it.n).affine_std == exp).all()
    hb = it.d_b.cpu().numpy().view(np.uint32).reshape(it.n, 16)
    hs = it.d_s.cpu().numpy().view(np.uint32).reshape(it.n, 8)
    with mh.MsmMulti(devices=[0, 0], host_threads=threads) as m:
        assert (m.msm(hb, hs, mh.FORM_MONT).affine_std == exp).all()
    if threads == 1:
        with pytest.raises(mh.MsmError) as e:
            mh.MsmContext(host_threads=65)
        assert e.value.code == mh.ERR_BAD_ARG


def test_one_rank_over_the_rccl_backend():
    """torch.distributed's `nccl` backend (= RCCL) had run on no hardware (VERDICT r4 missing 1: no box with two devices).  One torchrun rank on
    the one device there is: init_process_group("nccl", device_id), the 100-byte all_gather_into_tensor of mopro_msm_hip.distributed on the
    rank's device with its pinned staging and stream synchronisation, barrier, all_gather_object -- every call the N-rank path makes,
    on a communicator of one."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MSM_TEST_BACKEND="nccl", MSM_TEST_LOG_N="16", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_gpu_worker.py")]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert len(lines) == 1 and lines[0]["ok"] and lines[0]["backend"] == "nccl" and lines[0]["rccl_exchange_with_one_rank"] is True


def test_bench_gpus_2_starts_without_torchrun():
    """VERDICT r3 missing #1: `python bench.py --gpus N` started PLAINLY -- the way the driver starts its N = 1 line -- must launch its N
    ranks itself (a child `python -m torch.distributed.run ...`, never a re-exec), relay rank 0's JSON line and exit code.  On this
    1-GPU box both ranks share cuda:0 (--debug-same-device: exchange over gloo)."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--debug-same-device", "--steps", "2", "--warmup", "1",
                        "--log-n", "18", "--pre-warm-ms", "20", "--no-host-legs", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["bit_exact"] and j["exchange"]["world_seen"] == 2 and j["config"]["n_per_gpu"] == 1 << 17


def test_two_ranks_bad_scalar_on_one_rank_fails_everywhere():
    """NO RANK MAY HANG (metal_msm.rs:647-656 returns Err): rank 1's shard holds one scalar >= 2^254, so its local MSM fails with
    MSM_ERR_BAD_ARG.  It still joins the all-gather (identity + status word) and BOTH ranks raise MsmError(ERR_BAD_ARG) well
    inside the timeout; the healthy rank does not sit in the collective waiting for a partner that never comes."""
    p, lines = _torchrun_two_ranks({"MSM_TEST_BAD_SCALAR_RANK": "1"})
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert len(lines) == 2 and all(l["ok"] for l in lines), lines
    assert all(l["error_code"] == mh.ERR_BAD_ARG for l in lines), lines


def test_multi_bad_scalar_in_one_shard_fails_without_hanging(inst20):
    """msm_multi: a scalar >= 2^254 in ONE shard.  Host fold ({0,0,0}): the call returns MSM_ERR_BAD_ARG naming the rank.  RCCL
    exchange (forced on the single device {0}; on a multi-GPU host AUTO picks it): the ranks rendezvous on the host before the
    collective and all skip it -- the call returns the same code instead of waiting in ncclAllGather."""
    it = inst20
    n = 1 << 17
    hb = it.d_b[: n * 16].cpu().numpy().view(np.uint32).reshape(n, 16)
    hs = it.d_s[: n * 8].cpu().numpy().view(np.uint32).reshape(n, 8).copy()
    good, _ = _expected(it.dot(0, n))
    bad = hs.copy()
    bad[n // 2 + 5, 7] = 0x40000000  # 2^254: inside shard 1 of 3
    import torch
    d_bad = torch.from_numpy(bad.view(np.int32).reshape(-1)).to("cuda:0")
    with mh.MsmMulti(devices=[0, 0, 0]) as m:
        with pytest.raises(mh.MsmError) as e:
            m.msm(hb, bad, mh.FORM_MONT)
        assert e.value.code == mh.ERR_BAD_ARG and "rank 1" in str(e.value), str(e.value)
        cuts = [g * n // 3 for g in range(4)]
        with pytest.raises(mh.MsmError) as e:
            m.msm_device([it.d_b.data_ptr() + 64 * cuts[g] for g in range(3)], [d_bad.data_ptr() + 32 * cuts[g] for g in range(3)],
                         [cuts[g + 1] - cuts[g] for g in range(3)])
        assert e.value.code == mh.ERR_BAD_ARG
        assert (m.msm(hb, hs, mh.FORM_MONT).affine_std == good).all()  # the handle stays usable
        ex_ms, shard_ms = m.exchange_stats()
        assert len(shard_ms) == 3 and all(t > 0 for t in shard_ms) and ex_ms >= 0
    try:
        m = mh.MsmMulti(devices=[0], exchange=mh.EXCHANGE_RCCL)
    except mh.MsmError as err:
        assert err.code == mh.ERR_RCCL  # no librccl on this box
        return
    with m:
        with pytest.raises(mh.MsmError) as e:
            m.msm(hb, bad, mh.FORM_MONT)
        assert e.value.code == mh.ERR_BAD_ARG
        assert (m.msm(hb, hs, mh.FORM_MONT).affine_std == good).all()


def test_planner_choice_is_near_its_neighbours(hk):
    """the window table is measured, not modelled -- so compare on THIS box the planner's width with the
    neighbouring usable widths (resident call, median of 9), at an 8-GPU shard size, the BASELINE size and a size in the c = 17
    range.  A WALL-CLOCK comparison has no place in a pass/fail gate (one noisy neighbour on the box and `pytest -x` never reaches
    the files behind this one): the numbers go to gpurun_out/planner_neighbours.json and a ratio above 1.12 only warns."""
    import time
    import warnings
    report = {}
    for logn, neighbours in ((17, (13, 15)), (20, (15, 17)), (21, (16,))):
        it = Instance(hk, logn, seed=0xB25400A1 + logn)

        def med(ctx):
            for _ in range(3):
                ctx.msm_device(it.d_b.data_ptr(), it.d_s.data_ptr(), it.n)
            ts = []
            for _ in range(9):
                t = time.perf_counter()
                ctx.msm_device(it.d_b.data_ptr(), it.d_s.data_ptr(), it.n)
                ts.append(time.perf_counter() - t)
            return sorted(ts)[4]

        with mh.MsmContext() as c0:
            t_plan = med(c0)
        row = {"planner_ms": round(t_plan * 1e3, 4)}
        best = t_plan
        for c in neighbours:
            with mh.MsmContext(window_bits=c) as cx:
                t = med(cx)
                row["c%d_ms" % c] = round(t * 1e3, 4)
                best = min(best, t)
        row["ratio_to_best"] = round(t_plan / best, 4)
        report["2^%d" % logn] = row
        if t_plan > 1.12 * best:
            warnings.warn("planner width at 2^%d is %.1f %% slower than a neighbouring width on this box: %r" % (logn, 100 * (t_plan / best - 1), row))
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "planner_neighbours.json"), "w") as f:
            json.dump(report, f, indent=1)
    except OSError:
        pass


def test_trace_and_roctx_switches():
    """SURVEY section 5 observability: MSM_HIP_TRACE=1 prints one line per call (plan, path, per-stage device times),
    MSM_HIP_ROCTX=1 brackets the stages with roctx ranges (library dlopen'ed); both are read once per process."""
    code = (
        "import sys; sys.path[:0] = [%r, %r]\n"
        "import numpy as np, mopro_msm_hip as mh\n"
        "g = np.load(%r)\n"
        "with mh.MsmContext() as c:\n"
        "    r = c.msm(g['bases'], g['scalars'], mh.FORM_STD, g['inf'])\n"
        "    assert (r.affine_std == g['expected']).all()\n"
        "    print('stages', c.timings()['sort_ms'] > 0)\n"
    ) % (ROOT, os.path.join(ROOT, "gpu-acceleration_amd"), os.path.join(ROOT, "tests", "golden", "msm_rand_n4096.npz"))
    env = dict(os.environ, MSM_HIP_TRACE="1", MSM_HIP_ROCTX="1")
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "stages True" in p.stdout
    line = [l for l in p.stderr.splitlines() if l.startswith("[msm_hip] host single-shot")]
    assert line and " glv 1 " in line[0] and "accumulate" in line[0] and "adds" in line[0], p.stderr[-2000:]


def test_streamed_and_multi_randomised_fuzz():
    """randomised net under the shared-bucket streaming (k_accumulate<INTO>: bucket read-modify-write, once/mid/long lists per chunk,
    flag words that clean themselves between chunks) and under msm_multi's sharding: 36 random (size, chunk, window, digit form, GLV,
    infinity mask, scalar skew) cases against the oracle, each also as a 2- or 3-shard in-process multi call"""
    rng = np.random.default_rng(20261003)
    nmax = 30000
    k_all = orc.gen_scalars(777, nmax, nonzero=True)
    bases_all = orc.gen_bases_from_logs(k_all, orc.FORM_MONT)
    s_all = orc.gen_scalars(778, nmax)
    for it in range(36):
        lg = int(rng.choice([8, 9, 10, 11]))
        n = int(rng.integers(2 << lg, min(nmax, 9 << lg)))
        off = int(rng.integers(0, nmax - n + 1))
        bases, s = bases_all[off:off + n].copy(), s_all[off:off + n].copy()
        mode = int(rng.integers(0, 6))
        if mode == 1:
            s[:] = s[0]                      # one bucket per window holds everything: long buckets in every chunk
        elif mode == 2:
            s = s[np.arange(n) % 5]
        elif mode == 3:
            s[:, 1:] = 0                     # only the lowest windows are populated
        elif mode == 4:
            s[rng.random(n) < 0.7] = 0       # most digits skipped
        elif mode == 5 and n > 1:
            bases[1::2] = bases[0]           # P + P inside buckets and across chunks
        inf = (rng.random(n) < rng.choice([0.002, 0.2])).astype(np.uint8) if rng.random() < 0.4 else None
        wb = int(rng.choice([0, 0, 4, 7, 10, 13, 16]))
        flags = (mh.FLAG_UNSIGNED_DIGITS if rng.random() < 0.2 else 0) | (mh.FLAG_NO_GLV if rng.random() < 0.4 else 0)
        exp, einf, _ = orc.msm_pippenger(bases, s, orc.FORM_MONT, inf)
        tag = dict(case=it, n=n, chunk_log2=lg, mode=mode, wb=wb, flags=flags, inf=inf is not None)
        with mh.MsmContext(window_bits=wb, flags=flags, stream_chunk_log2=lg) as c:
            for rep in range(2):             # twice: the flag words and list counters must be clean again
                r = c.msm(bases, s, mh.FORM_MONT, inf)
                assert r.is_infinity == bool(einf) and (r.affine_std == exp).all(), (tag, rep)
            assert c.timings()["stream_chunks"] >= 2
        G = 2 + it % 2
        with mh.MsmMulti(devices=[0] * G, window_bits=wb, flags=flags, stream_chunk_log2=lg if it % 3 == 0 else 0) as m:
            r = m.msm(bases, s, mh.FORM_MONT, inf)
            assert r.is_infinity == bool(einf) and (r.affine_std == exp).all(), (tag, "multi", G)
