#!/usr/bin/env python3
"""Single-GPU device-call times at the shard sizes of the multi-GPU configurations (2^20 / G and 2^24 / G points, G = 1, 2, 4, 8):
what every rank of a G-GPU run would take alone.  Writes gpurun_out/shard_times.json -- committed as profiles/shard_times.json, from which
bench.py prints `projected_strong_scaling` (the 8-GPU run itself is the driver's).  Median of `--reps` calls on a ramped-up device.

    python tools/shard_times.py [--logs 17,18,19,20,21,22,23,24] [--reps 15] [--build "round 3 final"]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gpu-acceleration_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--logs", default="17,18,19,20,21,22,23,24")
    ap.add_argument("--reps", type=int, default=15)
    ap.add_argument("--build", default="round 3")
    args = ap.parse_args()
    import torch
    import mopro_msm_hip as mh
    from mopro_msm_hip import testhooks as th

    out = {"build": args.build, "what": "msm_bn254_g1_device, inputs resident in HBM, median ms of %d calls after a 150 ms ramp-up" % args.reps,
           "device_call_ms_by_log2_points": {}, "exchange_ms_estimate": 0.12,
           "exchange_note": "an ESTIMATE: the largest exchange measured end to end so far (two gloo ranks on one device, profiles/r4_final_bench_torchrun_2x_same_device.json); no multi-GPU run exists"}
    for lg in [int(x) for x in args.logs.split(",")]:
        n = 1 << lg
        d_b = torch.empty(n * 16, dtype=torch.int32, device="cuda:0")
        d_s = torch.empty(n * 8, dtype=torch.int32, device="cuda:0")
        with th.HooksContext() as gen:
            gen.generate_device(0xB2540500 + lg, 0xB2540600 + lg, n, d_b.data_ptr(), d_s.data_ptr())
        torch.cuda.synchronize()
        with mh.MsmContext(max_points=n) as ctx:
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.15:
                ctx.msm_device(d_b.data_ptr(), d_s.data_ptr(), n)
            ts = []
            for _ in range(args.reps):
                t = time.perf_counter()
                ctx.msm_device(d_b.data_ptr(), d_s.data_ptr(), n)
                ts.append((time.perf_counter() - t) * 1e3)
            ts.sort()
            out["device_call_ms_by_log2_points"][str(lg)] = round(ts[len(ts) // 2], 4)
            print("2^%d: median %.4f min %.4f ms" % (lg, ts[len(ts) // 2], ts[0]), flush=True)
        del d_b, d_s
        torch.cuda.empty_cache()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "shard_times.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
