#!/usr/bin/env python3
"""k_accumulate inside back-to-back msm_bn254_g1_device calls: with the base conversion on the copy stream beside the sort (default) and with
everything on one stream (stage timing on), 200 calls each, interleaved blocks; and the resident call (no conversion at all).  Separates
"the box sustains a lower rate" from "the concurrent conversion disturbs the accumulation".   usage: tools/sustained_probe.py [log_n]"""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gpu-acceleration_amd")]
import numpy as np, torch
import mopro_msm_hip as mh
from mopro_msm_hip import testhooks as th
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << lg
d_b = torch.empty(n * 16, dtype=torch.int32, device="cuda:0"); d_s = torch.empty(n * 8, dtype=torch.int32, device="cuda:0")
with th.HooksContext() as gen:
    gen.generate_device(71, 72, n, d_b.data_ptr(), d_s.data_ptr())
torch.cuda.synchronize()
hb = d_b.cpu().numpy().view(np.uint32).reshape(n, 16)
with mh.MsmContext() as c:
    c.set_kernel_timing(1)
    c.upload_bases(hb, mh.FORM_MONT)
    def block(kind, reps=200):
        c.set_stage_timing(kind == "one stream")
        c.reset_kernel_stats()
        t0 = time.perf_counter()
        for _ in range(reps):
            if kind == "resident": c.msm_resident_device(d_s.data_ptr(), n)
            else: c.msm_device(d_b.data_ptr(), d_s.data_ptr(), n)
        wall = (time.perf_counter() - t0) * 1e3 / reps
        avg, cnt = c.accumulate_kernel_stats()
        clk = c.clock_stats()
        c.set_stage_timing(False)
        return wall, avg, clk
    for _ in range(100): c.msm_device(d_b.data_ptr(), d_s.data_ptr(), n)
    for rnd in range(3):
        for kind in ("conversion beside the sort", "one stream", "resident"):
            wall, acc, clk = block(kind)
            print(f"round {rnd} {kind:28s}: call {wall:.4f} ms, k_accumulate {acc:.4f} ms (200 calls back to back); sclk {clk['sclk_ghz']:.3f} GHz, "
                  f"{clk['cycles_per_addition']:.0f} shader cycles per addition (first wavefront), kernel {acc * clk['sclk_ghz']:.4f} Mcycles", flush=True)
    # the same kernel in single calls separated by idle gaps: does the clock differ from the sustained loop's?
    for gap_ms in (0, 1, 5, 20, 100):
        c.reset_kernel_stats()
        for _ in range(20):
            c.msm_device(d_b.data_ptr(), d_s.data_ptr(), n)
            time.sleep(gap_ms * 1e-3)
        acc, _ = c.accumulate_kernel_stats()
        clk = c.clock_stats()
        print(f"20 calls with {gap_ms:3d} ms idle between them: k_accumulate {acc:.4f} ms, sclk {clk['sclk_ghz']:.3f} GHz, "
              f"{clk['cycles_per_addition']:.0f} cycles per addition, kernel {acc * clk['sclk_ghz']:.4f} Mcycles", flush=True)
