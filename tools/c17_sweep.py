"""window width 16 vs 17 (15 windows of 65536 buckets) from 2^20 points up, resident call, interleaved."""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gpu-acceleration_amd")]
import torch
import mopro_msm_hip as mh
from mopro_msm_hip import testhooks as th
gen = th.HooksContext()
for lg in [int(a) for a in sys.argv[1:]] or [20, 21, 22, 23, 24]:
    n = 1 << lg
    d_b = torch.empty(n * 16, dtype=torch.int32, device="cuda:0"); d_s = torch.empty(n * 8, dtype=torch.int32, device="cuda:0")
    gen.generate_device(1, 2, n, d_b.data_ptr(), d_s.data_ptr()); torch.cuda.synchronize()
    rows = {16: [], 17: []}
    with mh.MsmContext(window_bits=16) as c16, mh.MsmContext(window_bits=17) as c17:
        c16.set_kernel_timing(1); c17.set_kernel_timing(1)
        for rnd in range(3):
            for c, ctx in ((16, c16), (17, c17)):
                for _ in range(2): ctx.msm_device(d_b.data_ptr(), d_s.data_ptr(), n)
                ctx.reset_kernel_stats()
                ts = []
                for _ in range(10 if lg <= 22 else 4):
                    t = time.perf_counter(); ctx.msm_device(d_b.data_ptr(), d_s.data_ptr(), n); ts.append((time.perf_counter() - t) * 1e3)
                rows[c].append((statistics.median(ts), ctx.accumulate_kernel_stats()[0]))
    for c in (16, 17):
        print(f"2^{lg} c={c} median {statistics.median(x[0] for x in rows[c]):.3f} ms  k_accumulate {statistics.median(x[1] for x in rows[c]):.3f} ms", flush=True)
gen.close()
