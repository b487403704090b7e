"""Window-split knob (MSM_HIP_SPLIT_B = windows in the last group, 0 = no split) vs size, resident call, one process."""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gpu-acceleration_amd")]
import torch
import mopro_msm_hip as mh
from mopro_msm_hip import testhooks as th
sizes = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "16,17,18,19,20,21").split(",")]
Bs = (sys.argv[2] if len(sys.argv) > 2 else "0,1,2,3,4,5,6,auto").split(",")
os.environ["MSM_HIP_SPLIT_MIN_LOG2"] = "0"
gen = th.HooksContext()
for lg in sizes:
    n = 1 << lg
    d_b = torch.empty(n * 16, dtype=torch.int32, device="cuda:0"); d_s = torch.empty(n * 8, dtype=torch.int32, device="cuda:0")
    gen.generate_device(1, 2, n, d_b.data_ptr(), d_s.data_ptr()); torch.cuda.synchronize()
    with mh.MsmContext() as c:
        ref = None
        rows = {B: [] for B in Bs}
        for rnd in range(3):
            for B in Bs:
                if B == "auto": os.environ.pop("MSM_HIP_SPLIT_B", None)
                else: os.environ["MSM_HIP_SPLIT_B"] = B
                for _ in range(3): r = c.msm_device(d_b.data_ptr(), d_s.data_ptr(), n)
                ts = []
                for _ in range(25):
                    t = time.perf_counter(); r = c.msm_device(d_b.data_ptr(), d_s.data_ptr(), n); ts.append((time.perf_counter() - t) * 1e3)
                aff = r.affine_std.copy()
                if ref is None: ref = aff
                rows[B].append((statistics.median(ts), c.timings()["accumulate_windows"], bool((aff == ref).all())))
        os.environ.pop("MSM_HIP_SPLIT_B", None)
        for B in Bs:
            v = rows[B]
            print(f"2^{lg} B={B:>4s} median {statistics.median(x[0] for x in v):.4f} ms  timed windows {v[0][1]}  same={all(x[2] for x in v)}", flush=True)
gen.close()
