"""Two contexts used alternately from ONE thread (A, B, A, B ...): per-context stage times.  Is the second pipeline of the batch slower
in its memory-bound kernels because the two working sets evict each other from the Infinity Cache?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gpu-acceleration_amd")]
import torch
import mopro_msm_hip as mh
from mopro_msm_hip import testhooks as th
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << lg
gen = th.HooksContext()
d_b = torch.empty(n * 16, dtype=torch.int32, device="cuda:0"); d_s = torch.empty(n * 8, dtype=torch.int32, device="cuda:0")
gen.generate_device(1, 2, n, d_b.data_ptr(), d_s.data_ptr()); torch.cuda.synchronize()
A, B = mh.MsmContext(), mh.MsmContext()
for c in (A, B): c.set_stage_timing(True)
def run(seq, label):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.15:
        for c in seq: c.msm_device(d_b.data_ptr(), d_s.data_ptr(), n)
    acc = {id(A): [], id(B): []}
    for _ in range(10):
        for c in seq:
            c.msm_device(d_b.data_ptr(), d_s.data_ptr(), n)
            tm = c.timings()
            acc[id(c)].append((tm["decompose_ms"], tm["sort_ms"], tm["accumulate_ms"], tm["reduce_ms"], tm["total_ms"]))
    for name, c in (("A", A), ("B", B)):
        v = acc[id(c)]
        if v:
            m = [sorted(x[i] for x in v)[len(v) // 2] for i in range(5)]
            print(f"{label:12s} ctx {name}: decompose {m[0]:.3f} sort {m[1]:.3f} accumulate {m[2]:.3f} reduce {m[3]:.3f} total {m[4]:.3f}", flush=True)
run([A], "A alone")
run([B], "B alone")
run([A, B], "alternating")
run([A, A, B, B], "AABB")
