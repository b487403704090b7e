"""Row f3: arkworks compressed point images + the reference's instance-file harness (utils/preprocess.rs,
arkworks_pippenger.rs:44-80,155-177).  CPU part: oracle and host compressor against the golden images, file format
round trips.  GPU part: on-device square roots against the oracle, MSM from files."""
import json
import os
import struct

import numpy as np
import pytest

import mopro_msm_hip as mh
from mopro_msm_hip import instances as inst
from oracle import bn254_oracle as orc

GOLD = os.path.join(os.path.dirname(__file__), "golden", "compressed_points.json")


def _gold():
    g = json.load(open(GOLD))
    img = b"".join(bytes.fromhex(v["image"]) for v in g["valid"])
    xy = np.zeros((len(g["valid"]), 16), np.uint32)
    inf = np.zeros(len(g["valid"]), np.uint8)
    for i, v in enumerate(g["valid"]):
        xy[i, :8] = orc.int_to_words(int(v["x"], 16))
        xy[i, 8:] = orc.int_to_words(int(v["y"], 16))
        inf[i] = v["inf"]
    return g, img, xy, inf


def _to_mont(xy_std):
    out = np.zeros_like(xy_std)
    for i in range(xy_std.shape[0]):
        out[i, :8] = orc.fq_to_mont(xy_std[i, :8])
        out[i, 8:] = orc.fq_to_mont(xy_std[i, 8:])
    return out


# ---------------------------------------------------------------- CPU ----------
def test_oracle_decompress_matches_golden():
    g, img, xy, inf = _gold()
    oxy, oinf, bad = orc.g1_decompress(img, orc.FORM_STD)
    assert bad == -1
    assert (oinf == inf).all()
    assert (oxy[inf == 0] == xy[inf == 0]).all()
    assert orc.g1_compress(xy, orc.FORM_STD, inf) == img
    for k, v in enumerate(g["invalid"]):
        _, _, bad = orc.g1_decompress(img[:64] + bytes.fromhex(v["image"]) + img[:32])
        assert bad == 2, v["why"]


def test_host_compress_matches_golden_and_oracle():
    g, img, xy, inf = _gold()
    assert mh.compress_points(xy, mh.FORM_STD, inf) == img
    assert mh.compress_points(_to_mont(xy), mh.FORM_MONT, inf) == img
    k = orc.gen_scalars(77, 3000, nonzero=True)
    bases = orc.gen_bases_from_logs(k, orc.FORM_MONT)  # above the host compressor's threading threshold? no: 3000 < 8192
    assert mh.compress_points(bases, mh.FORM_MONT) == orc.g1_compress(bases, orc.FORM_MONT)
    k = orc.gen_scalars(78, 9000, nonzero=True)
    bases = orc.gen_bases_from_logs(k, orc.FORM_STD)   # threaded path
    assert mh.compress_points(bases, mh.FORM_STD) == orc.g1_compress(bases, orc.FORM_STD)
    with pytest.raises(mh.MsmError):
        mh.compress_points(np.zeros((0, 16), np.uint32))


def test_instance_files_round_trip(tmp_path):
    g, img, xy, inf = _gold()
    d = str(tmp_path / "16x3")
    sc = [orc.gen_scalars(5 + i, 64) for i in range(3)]
    for i in range(3):
        inst.serialize_input(d, img, sc[i], append=i != 0)
    # on-disk layout: u64 LE length + 32-byte records, instances appended (preprocess.rs:193-223)
    raw = open(os.path.join(d, "points"), "rb").read()
    assert len(raw) == 3 * (8 + 64 * 32) and struct.unpack("<Q", raw[:8])[0] == 64 and raw[8:8 + 64 * 32] == img
    raw = open(os.path.join(d, "scalars"), "rb").read()
    assert len(raw) == 3 * (8 + 64 * 32) and raw[8:8 + 64 * 32] == sc[0].tobytes()
    pts, scs = inst.deserialize_input(d)
    assert len(pts) == 3 and all(p == img for p in pts)
    assert all((a == b).all() for a, b in zip(scs, sc))
    it = inst.FileInputIterator.open(d)
    assert len(list(it)) == 3
    # not appending truncates (File::create)
    inst.serialize_input(d, img[:96], sc[0][:3], append=False)
    pts, scs = inst.deserialize_input(d)
    assert len(pts) == 1 and len(pts[0]) == 96 and scs[0].shape == (3, 8)


def test_instance_files_errors(tmp_path):
    with pytest.raises(inst.HarnessError, match="could not open file"):
        inst.FileInputIterator(str(tmp_path / "missing"))
    d = tmp_path / "empty"
    d.mkdir()
    (d / "points").write_bytes(b"")
    (d / "scalars").write_bytes(b"")
    with pytest.raises(inst.HarnessError, match="failed to read at least one instance"):
        inst.FileInputIterator(str(d))
    # a truncated trailing instance ends the iteration instead of failing (preprocess.rs:117-127)
    g, img, xy, inf = _gold()
    d2 = str(tmp_path / "trunc")
    inst.serialize_input(d2, img, orc.gen_scalars(1, 64), append=False)
    with open(os.path.join(d2, "points"), "ab") as f:
        f.write(struct.pack("<Q", 10) + img[:100])
    with open(os.path.join(d2, "scalars"), "ab") as f:
        f.write(struct.pack("<Q", 10) + bytes(320))
    assert len(list(inst.FileInputIterator(d2))) == 1


def test_benchmark_result_csv(tmp_path):
    r = [inst.BenchmarkResult(16, 10, 1.25), inst.BenchmarkResult(20, 10, 2.0)]
    p = tmp_path / "hip_benchmark.txt"
    inst.write_csv(str(p), r)
    assert p.read_text().splitlines() == ["msm_size,num_msm,avg_processing_time(ms)", "16,10,1.25", "20,10,2.0"]


# ---------------------------------------------------------------- GPU ----------
@pytest.fixture(scope="module")
def ctx():
    c = mh.MsmContext()
    yield c
    c.close()


@pytest.mark.gpu
def test_gpu_decompress_golden(ctx):
    g, img, xy, inf = _gold()
    dxy, dinf = ctx.decompress(img)
    assert (dinf == inf).all()
    assert (dxy[inf == 0] == _to_mont(xy)[inf == 0]).all()
    for k, v in enumerate(g["invalid"]):
        with pytest.raises(mh.MsmError) as e:
            ctx.decompress(img[:96] + bytes.fromhex(v["image"]) + img[:32] + bytes.fromhex(v["image"]))
        assert e.value.code == mh.ERR_INVALID_DATA and e.value.first_invalid == 3, v["why"]
        with pytest.raises(mh.MsmError) as e:
            ctx.upload_compressed(bytes.fromhex(v["image"]))
        assert e.value.code == mh.ERR_INVALID_DATA and e.value.first_invalid == 0
    with pytest.raises(mh.MsmError) as e:
        ctx.decompress(b"")
    assert e.value.code == mh.ERR_EMPTY
    with pytest.raises(mh.MsmError) as e:
        ctx.decompress(img[:33])
    assert e.value.code == mh.ERR_BAD_ARG
    # a failed upload leaves no resident set behind
    with pytest.raises(mh.MsmError) as e:
        ctx.msm_resident(orc.gen_scalars(1, 4))
    assert e.value.code == mh.ERR_STATE


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 63, 5000, 1 << 16])
def test_gpu_decompress_random_vs_oracle(ctx, n):
    k = orc.gen_scalars(1000 + n, n, nonzero=True)
    bases = orc.gen_bases_from_logs(k, orc.FORM_MONT)
    inf = np.zeros(n, np.uint8)
    inf[::97] = 1
    img = mh.compress_points(bases, mh.FORM_MONT, inf)
    dxy, dinf = ctx.decompress(img)
    oxy, oinf, bad = orc.g1_decompress(img, orc.FORM_MONT)
    assert bad == -1
    assert (dinf == oinf).all() and (dinf == inf).all()
    assert (dxy == oxy).all()
    assert (dxy[inf == 0] == bases[inf == 0]).all()
    # negated points: the other root
    neg = bases.copy()
    for i in range(min(n, 50)):
        neg[i, 8:] = orc.fq_sub(np.zeros(8, np.uint32), bases[i, 8:])
    dxy2, _ = ctx.decompress(mh.compress_points(neg[:50], mh.FORM_MONT))
    assert (dxy2 == neg[:min(n, 50)]).all()


@pytest.mark.gpu
def test_gpu_msm_from_compressed_equals_msm_from_coordinates(ctx):
    n = 20000
    k = orc.gen_scalars(31, n, nonzero=True)
    bases = orc.gen_bases_from_logs(k, orc.FORM_MONT)
    inf = np.zeros(n, np.uint8)
    inf[5::1000] = 1
    s = orc.gen_scalars(32, n)
    want = ctx.msm(bases, s, mh.FORM_MONT, inf)
    ctx.upload_compressed(mh.compress_points(bases, mh.FORM_MONT, inf))
    got = ctx.msm_resident(s)
    assert (got.affine_std == want.affine_std).all() and got.is_infinity == want.is_infinity
    aff, oi, _ = orc.msm_pippenger(bases, s, orc.FORM_MONT, inf)
    assert (got.affine_std == aff).all()
    # fewer scalars than resident points: truncates (metal_msm.rs:652-656)
    got = ctx.msm_resident(s[:777])
    aff, oi, _ = orc.msm_pippenger(bases[:777], s[:777], orc.FORM_MONT, inf[:777])
    assert (got.affine_std == aff).all()


@pytest.mark.gpu
def test_gpu_run_benchmark_from_instance_files(ctx, tmp_path):
    d = str(tmp_path / "vectors" / "10x3")
    res = inst.run_benchmark(10, 3, d, ctx)          # generates the vectors, then times them
    assert res.instance_size == 10 and res.num_instance == 3 and res.avg_processing_time > 0
    res2 = inst.run_benchmark(10, 3, d, ctx)         # "Vectors already generated"
    assert res2.num_instance == 3
    results = []
    durs = inst.benchmark_msm(inst.FileInputIterator(d), 2, ctx, results)
    assert len(durs) == 3 and len(results) == 3
    pts, scs = inst.deserialize_input(d)
    for images, scalars, r in zip(pts, scs, results):
        xy, oinf, bad = orc.g1_decompress(images, orc.FORM_MONT)
        assert bad == -1
        aff, oi, _ = orc.msm_pippenger(xy, scalars, orc.FORM_MONT, oinf)
        assert (r.affine_std == aff).all() and r.is_infinity == bool(oi)
    assert pts[0] != pts[1]
    inst.write_csv(str(tmp_path / "hip_benchmark.txt"), [res, res2])
    lines = (tmp_path / "hip_benchmark.txt").read_text().splitlines()
    assert lines[0] == inst.CSV_HEADER and lines[1].startswith("10,3,")
