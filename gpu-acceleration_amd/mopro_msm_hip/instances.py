"""Instance files and the benchmark harness of the reference, on the HIP engine.

Mirrors mopro-msm/src/msm/utils/preprocess.rs and the `run_benchmark` harnesses (arkworks_pippenger.rs:44-80,
155-177 and their siblings): an instance directory holds two files,

    points   = Vec<G1Affine>::serialize_compressed, appended once per instance   (preprocess.rs:193-223)
    scalars  = Vec<BigInt<4>>::serialize_compressed, appended once per instance

i.e. per instance a u64 little-endian length followed by 32-byte records: a compressed point image (x in standard form,
bit 255 = y > p - y, bit 254 = infinity) or a scalar in standard form.  Reading `points` is where arkworks spends a
square root per point on the CPU; here the images go to the GPU as they are (`MsmContext.upload_compressed`).

Same names as the reference: serialize_input / deserialize_input / FileInputIterator / gen_vectors /
benchmark_msm / run_benchmark / BenchmarkResult; the CSV line format is the reference's
"msm_size,num_msm,avg_processing_time(ms)" (arkworks_pippenger.rs:163).
"""
import os
import struct
import time
from dataclasses import dataclass

import numpy as np

from . import FORM_MONT, MsmContext, MsmError, compress_points, default_context

POINTS_FILE, SCALARS_FILE = "points", "scalars"
CSV_HEADER = "msm_size,num_msm,avg_processing_time(ms)"


class HarnessError(RuntimeError):
    """preprocess.rs:9-19 (SerializationError / FileOpenError / DeserializationError)."""


def serialize_input(dir, points_images, scalars, append):
    """preprocess.rs:193-223.  points_images: n x 32 bytes of compressed images (see compress_points);
    scalars: n x 8 standard-form words."""
    os.makedirs(dir, exist_ok=True)
    images = bytes(points_images)
    scalars = np.ascontiguousarray(scalars, dtype=np.uint32).reshape(-1, 8)
    if len(images) % 32:
        raise HarnessError("could not serialize: point images are 32 bytes each")
    mode = "ab" if append else "wb"
    with open(os.path.join(dir, POINTS_FILE), mode) as f:
        f.write(struct.pack("<Q", len(images) // 32))
        f.write(images)
    with open(os.path.join(dir, SCALARS_FILE), mode) as f:
        f.write(struct.pack("<Q", scalars.shape[0]))
        f.write(scalars.tobytes())


def _read_vec(f):
    head = f.read(8)
    if len(head) < 8:
        return None
    (n,) = struct.unpack("<Q", head)
    body = f.read(32 * n)
    if len(body) < 32 * n:
        return None  # arkworks: deserialisation error -> the iterator ends (preprocess.rs:117-127)
    return body


class FileInputIterator:
    """preprocess.rs:27-131: yields (point images: bytes, scalars: ndarray n x 8) per instance until either file ends."""

    def __init__(self, dir):
        try:
            self._pf = open(os.path.join(dir, POINTS_FILE), "rb")
            self._sf = open(os.path.join(dir, SCALARS_FILE), "rb")
        except OSError as e:
            raise HarnessError("could not open file: %s" % e)
        self._cached = self._next()
        if self._cached is None:
            self.close()
            raise HarnessError("failed to read at least one instance from file")

    @classmethod
    def open(cls, dir):
        return cls(dir)

    def _next(self):
        p = _read_vec(self._pf)
        if p is None:
            return None
        s = _read_vec(self._sf)
        if s is None:
            return None
        return p, np.frombuffer(s, dtype=np.uint32).reshape(-1, 8)

    def __iter__(self):
        return self

    def __next__(self):
        if self._cached is not None:
            item, self._cached = self._cached, None
        else:
            item = self._next()
        if item is None:
            self.close()
            raise StopIteration
        return item

    def close(self):
        self._pf.close()
        self._sf.close()


def deserialize_input(dir):
    """preprocess.rs:225-256: every instance of the directory as two lists."""
    pts, scs = [], []
    for p, s in FileInputIterator(dir):
        pts.append(p)
        scs.append(s)
    return pts, scs


def gen_vectors(instance_size, num_instance, dir, ctx=None, seed=0xB2540003):
    """preprocess.rs:180-191: `num_instance` instances of 2^instance_size random points and scalars.
    Points are k_i*G from the on-device generator of the hooks build (testhooks.HooksContext; the reference draws
    GAffine::rand from thread_rng).  `ctx` may be a HooksContext to reuse; a product MsmContext has no generator."""
    import torch

    from .testhooks import HooksContext

    gen = ctx if isinstance(ctx, HooksContext) else HooksContext()
    n = 1 << instance_size
    d_b = torch.empty(n * 16, dtype=torch.int32, device="cuda")
    d_s = torch.empty(n * 8, dtype=torch.int32, device="cuda")
    try:
        for i in range(num_instance):
            gen.generate_device((seed + 2 * i) & (2 ** 64 - 1), (seed + 2 * i + 1) & (2 ** 64 - 1), n, d_b.data_ptr(), d_s.data_ptr())
            torch.cuda.synchronize()
            bases = d_b.cpu().numpy().view(np.uint32).reshape(n, 16)
            scalars = d_s.cpu().numpy().view(np.uint32).reshape(n, 8)
            serialize_input(dir, compress_points(bases, FORM_MONT), scalars, append=i != 0)
    finally:
        if gen is not ctx:
            gen.close()


@dataclass
class BenchmarkResult:
    """utils/benchmark.rs:1-6"""
    instance_size: int
    num_instance: int
    avg_processing_time: float  # ms

    def csv_line(self):
        return "%d,%d,%s" % (self.instance_size, self.num_instance, repr(float(self.avg_processing_time)))


def benchmark_msm(instances, iterations=1, ctx=None, results=None):
    """arkworks_pippenger.rs:7-43 with the MSM swapped: per instance the average wall time of `iterations` MSMs.
    As in the reference the timed region starts with the instance already deserialised in memory (there: Vec<G1Affine>;
    here: bases decoded into HBM by upload_compressed) and covers the whole call on the scalars."""
    ctx = ctx or default_context()
    out = []
    for images, scalars in instances:
        ctx.upload_compressed(images)
        total = 0.0
        for _ in range(iterations):
            t0 = time.perf_counter()
            r = ctx.msm_resident(scalars)
            total += time.perf_counter() - t0
        if results is not None:
            results.append(r)
        out.append(total / iterations)
    return out


def run_benchmark(instance_size, num_instance, utils_dir, ctx=None):
    """arkworks_pippenger.rs:44-80: generate the vectors if the directory has none, then time every instance once."""
    try:
        FileInputIterator(utils_dir).close()
    except HarnessError:
        gen_vectors(instance_size, num_instance, utils_dir, ctx)
    durations = benchmark_msm(FileInputIterator(utils_dir), 1, ctx)
    return BenchmarkResult(instance_size, num_instance, sum(d * 1e3 for d in durations) / len(durations))


def write_csv(path, results):
    """arkworks_pippenger.rs:155-177 (test_run_multi_benchmarks)."""
    with open(path, "w") as f:
        f.write(CSV_HEADER + "\n")
        for r in results:
            f.write(r.csv_line() + "\n")


__all__ = ["serialize_input", "deserialize_input", "FileInputIterator", "gen_vectors", "benchmark_msm", "run_benchmark",
           "BenchmarkResult", "write_csv", "HarnessError", "CSV_HEADER", "MsmContext", "MsmError"]
