#!/usr/bin/env python3
"""Chunk schedules of the streamed host-pointer call (round 6): does a SMALL FIRST chunk (compute starts earlier) pay for a small LAST one
(less work after the last byte)?  MSM_HIP_STREAM_SCHEDULE (hooks build) sets the chunks as log2 sizes.  Pinned caller memory, N = 2^20 (and 2^21)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("MSM_HIP_LIB", os.path.join(ROOT, "gpu-acceleration_amd", "libmsm_hip_hooks.so"))  # the knob is read by the hooks build only
sys.path[:0] = [ROOT, os.path.join(ROOT, "gpu-acceleration_amd")]
import numpy as np, torch
import mopro_msm_hip as mh
from mopro_msm_hip import testhooks as th

SCHEDULES = {
    20: ["18,18,18,18", "17,18,18,18,17", "17,17,18,18,18", "18,18,18,17,17", "17,18,18,17,17,17", "16,17,18,18,18,16,16", "17,18,18,18,16,16", "19,18,18", "17,18,19,17,17",
         "16,16,17,18,18,18", "17,17,18,18,17,17"],
    21: ["19,19,19,19", "18,19,19,19,18", "18,18,19,19,18,18", "18,19,19,19,17,17", "17,18,19,19,19,17"],
}


def timed(fn, reps=9):
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < 0.15:
        fn()
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); fn(); ts.append((time.perf_counter() - t) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def main():
    gen = th.HooksContext()
    for lg in [int(a) for a in sys.argv[1:]] or [20, 21]:
        n = 1 << lg
        d_b = torch.empty(n * 16, dtype=torch.int32, device="cuda:0"); d_s = torch.empty(n * 8, dtype=torch.int32, device="cuda:0")
        gen.generate_device(1, 2, n, d_b.data_ptr(), d_s.data_ptr()); torch.cuda.synchronize()
        pb, ps = d_b.cpu().pin_memory(), d_s.cpu().pin_memory()
        hb, hs = pb.numpy().view(np.uint32).reshape(n, 16), ps.numpy().view(np.uint32).reshape(n, 8)
        ref = None
        for rnd in range(2):  # two rounds, interleaved: the box's clock drifts
            for sched in SCHEDULES[lg]:
                os.environ["MSM_HIP_STREAM_SCHEDULE"] = sched
                try:
                    with mh.MsmContext() as c:
                        med, mn = timed(lambda: c.msm(hb, hs, mh.FORM_MONT))
                        r = c.msm(hb, hs, mh.FORM_MONT)
                        tm = c.timings()
                finally:
                    os.environ.pop("MSM_HIP_STREAM_SCHEDULE", None)
                if ref is None:
                    ref = r.affine_std.copy()
                ok = bool((r.affine_std == ref).all())
                print(f"2^{lg} pinned round {rnd} schedule {sched:28s} median {med:7.3f} min {mn:7.3f} ms  chunks {tm['stream_chunks']} same_bits {ok}", flush=True)


if __name__ == "__main__":
    main()
