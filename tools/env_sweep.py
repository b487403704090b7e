"""Interleaved A/B of one per-call environment knob on the resident call, one process: tools/env_sweep.py NAME v1,v2,... [log_n,...]
('-' = unset).  Prints median latency and mean k_accumulate time per (size, value)."""
import os, sys, time, statistics
os.environ.setdefault("MSM_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpu-acceleration_amd", "libmsm_hip_hooks.so"))  # the A/B knobs this script sets are read by the HOOKS build only (round 5)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gpu-acceleration_amd")]
import torch
import mopro_msm_hip as mh
from mopro_msm_hip import testhooks as th
name = sys.argv[1]
vals = sys.argv[2].split(",")
sizes = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "16,17,18,19,20").split(",")]
gen = th.HooksContext()
for lg in sizes:
    n = 1 << lg
    d_b = torch.empty(n * 16, dtype=torch.int32, device="cuda:0"); d_s = torch.empty(n * 8, dtype=torch.int32, device="cuda:0")
    gen.generate_device(1, 2, n, d_b.data_ptr(), d_s.data_ptr()); torch.cuda.synchronize()
    with mh.MsmContext() as c:
        c.set_kernel_timing(1)
        ref = None
        rows = {v: [] for v in vals}
        for rnd in range(3):
            for v in vals:
                if v == "-": os.environ.pop(name, None)
                else: os.environ[name] = v
                for _ in range(3): r = c.msm_device(d_b.data_ptr(), d_s.data_ptr(), n)
                c.reset_kernel_stats()
                ts = []
                for _ in range(25):
                    t = time.perf_counter(); r = c.msm_device(d_b.data_ptr(), d_s.data_ptr(), n); ts.append((time.perf_counter() - t) * 1e3)
                acc, _ = c.accumulate_kernel_stats()
                aff = r.affine_std.copy()
                if ref is None: ref = aff
                rows[v].append((statistics.median(ts), acc, bool((aff == ref).all())))
        os.environ.pop(name, None)
        for v in vals:
            x = rows[v]
            print(f"2^{lg} {name}={v:>4s} median {statistics.median(y[0] for y in x):.4f} ms  k_accumulate {statistics.median(y[1] for y in x):.4f} ms  same={all(y[2] for y in x)}", flush=True)
gen.close()
