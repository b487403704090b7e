#!/usr/bin/env python3
"""Row f4: the window table of a resident base set with SHARED bucket arrays, swept over window width c
and table factor f.  For every (log n, c, f): upload time, ms per MSM of single resident calls and of the resident batch (two in
flight), stage times of one diagnostic call, and bit-exactness against the plain resident path and the closed form.

    python tools/table_sweep.py [--logn 18,20,22] [--configs plain,16x16,17x15,19x14,20x13] [--batch 8] [--reps 3]

A config is `plain` (no table), `CxF` (c bits, factor f; f must divide the number of windows) or `C` (f = all windows).
Knobs are read when a context is created, so every config gets a fresh context."""
import argparse
import os
os.environ.setdefault("MSM_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpu-acceleration_amd", "libmsm_hip_hooks.so"))  # the A/B knobs this script sets are read by the HOOKS build only (round 5)
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gpu-acceleration_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--logn", default="18,20,22")
    ap.add_argument("--configs", default="plain,16,17,19,20")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--no-glv", action="store_true")
    args = ap.parse_args()

    import torch
    import mopro_msm_hip as mh
    from mopro_msm_hip import testhooks as th
    from oracle import bn254_oracle as orc

    for logn in [int(x) for x in args.logn.split(",")]:
        n = 1 << logn
        seed = 0xB2540300 + logn
        d_b = torch.empty(n * 16, dtype=torch.int32, device="cuda:0")
        d_s = torch.empty(n * 8, dtype=torch.int32, device="cuda:0")
        with th.HooksContext() as gen:
            gen.generate_device(seed, seed + 1, n, d_b.data_ptr(), d_s.data_ptr())
        torch.cuda.synchronize()
        hb = d_b.cpu().pin_memory().numpy().view(np.uint32).reshape(n, 16)
        vecs = [torch.from_numpy(th.generate_scalars_host(seed + 1 + j, n).view(np.int32)).pin_memory().numpy().view(np.uint32).reshape(n, 8)
                for j in range(2)]
        k = th.generate_scalars_host(seed, n, nonzero=True)
        exp = [orc.closed_form_expected(k, v)[0] for v in vecs]
        ref = None
        for cfg in args.configs.split(","):
            for v in ("MSM_HIP_TABLE_C", "MSM_HIP_TABLE_F", "MSM_HIP_TABLE_GLV_MAX_LOG2"):
                os.environ.pop(v, None)
            flags = mh.FLAG_NO_GLV if args.no_glv else 0
            if cfg != "plain":
                flags |= mh.FLAG_WINDOW_TABLE
                if cfg.endswith("g"):  # `16g`: with the GLV split whatever the size
                    cfg_c = cfg[:-1]
                    os.environ["MSM_HIP_TABLE_GLV_MAX_LOG2"] = "23"
                else:
                    cfg_c = cfg
                c, _, f = cfg_c.partition("x")
                os.environ["MSM_HIP_TABLE_C"] = c
                if f:
                    os.environ["MSM_HIP_TABLE_F"] = f
            try:
                pl = mh.plan(n, 0, flags)
                with mh.MsmContext(flags=flags) as ctx:
                    t0 = time.perf_counter()
                    ctx.upload_bases(hb, mh.FORM_MONT)
                    up_ms = (time.perf_counter() - t0) * 1e3
                    r0 = [ctx.msm_resident(v) for v in vecs]
                    ok = all((r.affine_std == e).all() for r, e in zip(r0, exp))
                    if ref is None:
                        ref = [r.affine_std.copy() for r in r0]
                    ok = ok and all((r.affine_std == q).all() for r, q in zip(r0, ref))
                    # single calls
                    tw = time.perf_counter()
                    while time.perf_counter() - tw < 0.15:
                        ctx.msm_resident(vecs[0])
                    best_single = 1e9
                    for _ in range(args.reps):
                        t0 = time.perf_counter()
                        for j in range(args.batch):
                            ctx.msm_resident(vecs[j & 1])
                        best_single = min(best_single, (time.perf_counter() - t0) * 1e3 / args.batch)
                    # batch
                    bv = [vecs[j & 1] for j in range(args.batch)]
                    rb = ctx.msm_resident_batch(bv, want_affine=True)
                    ok = ok and all((r.affine_std == exp[j & 1]).all() for j, r in enumerate(rb))
                    best_batch = 1e9
                    for _ in range(args.reps):
                        t0 = time.perf_counter()
                        ctx.msm_resident_batch(bv, want_affine=False)
                        best_batch = min(best_batch, (time.perf_counter() - t0) * 1e3 / args.batch)
                    ctx.set_stage_timing(True)
                    ctx.msm_resident(vecs[0])
                    tm = ctx.timings()
                    ctx.set_stage_timing(False)
                print("2^%d %-7s c %2d W %2d f %2d arrays %2d nb 2^%d glv %d table %7.1f MB upload %7.1f ms | single %.4f batch %.4f ms/MSM | "
                      "decomp %.3f sort %.3f acc %.3f reduce %.3f finish %.3f adds %d | exact=%s"
                      % (logn, cfg, pl.window_bits, pl.num_windows, pl.table_factor, pl.bucket_arrays, pl.num_buckets.bit_length() - 1, pl.glv,
                         pl.table_bytes / 1e6, up_ms, best_single, best_batch, tm["decompose_ms"], tm["sort_ms"], tm["accumulate_ms"],
                         tm["reduce_ms"], tm["finish_ms"], tm["num_adds"], ok), flush=True)
            except mh.MsmError as e:
                print("2^%d %-7s FAILED: %s (%d)" % (logn, cfg, e, e.code), flush=True)
        del d_b, d_s
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
