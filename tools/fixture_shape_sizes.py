#!/usr/bin/env python3
"""Round 6 (NOTES_r6 section 16): the reference's fixture shape (one (base, scalar) sequence repeated T times, metal_msm.rs:706-730) over SIZES -- T = 1 (uniform), 8, 32,
128 at 2^16 ... 2^20 points through the device call: ms per MSM after the clock ramp, stage times and the list counters of k_combine_pieces (hooks build)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gpu-acceleration_amd")]
import numpy as np, torch
from mopro_msm_hip import testhooks as th
def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int32).reshape(-1).copy()).cuda()
with th.HooksContext() as c:
    for lg in (16, 17, 18, 19, 20):
        n = 1 << lg
        d_b = torch.empty(n * 16, dtype=torch.int32, device="cuda:0"); d_s = torch.empty(n * 8, dtype=torch.int32, device="cuda:0")
        c.generate_device(1, 2, n, d_b.data_ptr(), d_s.data_ptr()); torch.cuda.synchronize()
        s = d_s.cpu().numpy().view(np.uint32).reshape(n, 8); b = d_b.cpu().numpy().view(np.uint32).reshape(n, 16)
        for T in (1, 8, 32, 128):
            L = n // T
            tb, ts_ = dev(np.tile(b[:L], (T, 1))), dev(np.tile(s[:L], (T, 1)))
            t_w = time.perf_counter()
            while time.perf_counter() - t_w < 0.1: c.msm_device(tb.data_ptr(), ts_.data_ptr(), n)
            ts = []
            for _ in range(15):
                t0 = time.perf_counter(); c.msm_device(tb.data_ptr(), ts_.data_ptr(), n); ts.append((time.perf_counter() - t0) * 1e3)
            c.set_stage_timing(True)
            c.msm_device(tb.data_ptr(), ts_.data_ptr(), n); c.msm_device(tb.data_ptr(), ts_.data_ptr(), n)
            t = c.timings(); c.set_stage_timing(False)
            st = {k[:-3]: round(v, 3) for k, v in t.items() if k.endswith("_ms") and v > 0.004 and k != "total_ms"}
            print(f"2^{lg} T={T:3d}  {sorted(ts)[7]:.3f} ms  {st}  {c.list_counts()}", flush=True)
