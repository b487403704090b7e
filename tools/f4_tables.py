"""SURVEY section 8 row f4 / VERDICT r1 item 6(ii): precomputed 2-window table for resident bases.

A 2-window table stores, next to every base P_i, a second point T_i so that the scalar's upper half can be accumulated in the SAME
windows as its lower half: 2N base records, half the windows, the same number of mixed additions.  The GLV split (csrc/glv_bn254.hpp) IS
that table with T_i = phi(P_i) = lambda * P_i (and 127-bit halves): the engine builds it in k_convert_bases, keeps it resident
(msm_bn254_g1_upload_bases) and already runs on it up to 2^19 points.  So the question "does a 2-window table pay at 2^20 / 2^22" is
measured here as: the same instance with the table forced on (MSM_HIP_GLV_MAX_LOG2=23) and off, for the window widths the table makes
affordable (c = 16, 17, 18: fewer windows, more buckets), k_accumulate and end-to-end, interleaved A/B, bases and scalars in HBM.
usage: python tools/f4_tables.py [log_n ...]"""
import os, sys, time, statistics
os.environ.setdefault("MSM_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpu-acceleration_amd", "libmsm_hip_hooks.so"))  # the A/B knobs this script sets are read by the HOOKS build only (round 5)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gpu-acceleration_amd")]
import torch
import mopro_msm_hip as mh
from mopro_msm_hip import testhooks as th

sizes = [int(a) for a in sys.argv[1:]] or [20, 22]
gen = th.HooksContext()
for lg in sizes:
    n = 1 << lg
    d_b = torch.empty(n * 16, dtype=torch.int32, device="cuda:0"); d_s = torch.empty(n * 8, dtype=torch.int32, device="cuda:0")
    gen.generate_device(1, 2, n, d_b.data_ptr(), d_s.data_ptr()); torch.cuda.synchronize()
    configs = [("no table", False, c) for c in (16, 17)] + [("2-window table", True, c) for c in (16, 17, 18)]
    rows = {k: [] for k in configs}
    ref = None
    for rnd in range(3):
        for cfg in configs:
            label, table, c = cfg
            os.environ["MSM_HIP_GLV_MAX_LOG2"] = "23" if table else "0"
            with mh.MsmContext(window_bits=c) as ctx:
                ctx.set_kernel_timing(1)
                for _ in range(3): r = ctx.msm_device(d_b.data_ptr(), d_s.data_ptr(), n)
                ctx.reset_kernel_stats()
                ts = []
                for _ in range(15):
                    t = time.perf_counter(); r = ctx.msm_device(d_b.data_ptr(), d_s.data_ptr(), n); ts.append((time.perf_counter() - t) * 1e3)
                acc, _ = ctx.accumulate_kernel_stats()
                p = mh.plan(n, c)
                aff = r.affine_std.copy()
                if ref is None: ref = aff
                rows[cfg].append((statistics.median(ts), acc, p.num_windows, p.num_buckets, int(p.virtual_points), bool((aff == ref).all())))
    os.environ.pop("MSM_HIP_GLV_MAX_LOG2", None)
    base = statistics.median(x[0] for x in rows[configs[0]])
    for cfg in configs:
        v = rows[cfg]
        med, acc = statistics.median(x[0] for x in v), statistics.median(x[1] for x in v)
        print(f"2^{lg} {cfg[0]:15s} c={cfg[2]:2d}  W={v[0][2]:2d} x {v[0][3]:6d} buckets, {v[0][4]:8d} records gathered from: "
              f"end to end {med:7.3f} ms ({(med / base - 1) * 100:+5.1f} %)  k_accumulate {acc:7.3f} ms  same result={all(x[5] for x in v)}", flush=True)
gen.close()
