#!/usr/bin/env python3
"""Is k_accumulate clock/power limited under sustained load?  The same 2^20 MSM (device call) repeated with an idle gap of g ms between
calls: hipEvent time of k_accumulate per gap, and the multiplier calibration (k_calibrate, ~1 ms bursts) right after a sustained run
vs after idling.  (Observed in round 3: an accumulate launch that followed 1.4 ms of light sort work ran 37 % faster than in steady
state; boxes differ by 8 % in k_accumulate at the same calibrated multiplier rate.)   python tools/throttle_probe.py [log_n]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gpu-acceleration_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    import torch
    import mopro_msm_hip as mh
    from mopro_msm_hip import testhooks as th

    lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    n = 1 << lg
    d_b = torch.empty(n * 16, dtype=torch.int32, device="cuda:0")
    d_s = torch.empty(n * 8, dtype=torch.int32, device="cuda:0")
    with th.HooksContext() as hk, mh.MsmContext(max_points=n) as ctx:
        ctx.set_kernel_timing(1)
        hk.generate_device(11, 12, n, d_b.data_ptr(), d_s.data_ptr())
        torch.cuda.synchronize()
        for gap_ms in (0.0, 0.2, 0.5, 1.0, 2.0, 5.0, 0.0):
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.15:  # settle into this duty cycle
                ctx.msm_device(d_b.data_ptr(), d_s.data_ptr(), n)
                if gap_ms:
                    time.sleep(gap_ms * 1e-3)
            ctx.reset_kernel_stats()
            calls, t0 = 0, time.perf_counter()
            wall = []
            while time.perf_counter() - t0 < 0.25:
                t = time.perf_counter()
                ctx.msm_device(d_b.data_ptr(), d_s.data_ptr(), n)
                wall.append((time.perf_counter() - t) * 1e3)
                calls += 1
                if gap_ms:
                    time.sleep(gap_ms * 1e-3)
            acc, _ = ctx.accumulate_kernel_stats()
            wall.sort()
            m, f = hk.calibrate()  # right after this duty cycle
            print("gap %4.1f ms: k_accumulate %.4f ms, call median %.4f ms (%d calls), calibration right after: %.1f G mad/s %.1f G fpmul/s"
                  % (gap_ms, acc, wall[len(wall) // 2], calls, m / 1e9, f / 1e9), flush=True)
        time.sleep(0.5)
        m, f = hk.calibrate()
        print("after 0.5 s idle: calibration %.1f G mad/s %.1f G fpmul/s (cold clock)" % (m / 1e9, f / 1e9))
        best = (0, 0)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.3:
            m, f = hk.calibrate()
            best = (max(best[0], m), max(best[1], f))
        print("calibration back to back for 0.3 s: best %.1f G mad/s %.1f G fpmul/s, last %.1f / %.1f" % (best[0] / 1e9, best[1] / 1e9, m / 1e9, f / 1e9))


if __name__ == "__main__":
    main()
