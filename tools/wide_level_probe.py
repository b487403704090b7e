#!/usr/bin/env python3
"""Where one level of the bucket reduction's dependent chain spends its time.

Runs the hooks build's probes (include/msm_hip_testhooks.h: msm_probe_wide_level, msm_probe_launch_chain) on cuda:0 and prints
  * one eight-lane addition level in LDS (the body of lds_tree_wide), per part, in shader cycles and microseconds, for a lone
    wavefront (8 additions), 4 wavefronts on 4 SIMDs (32 additions) and 8 wavefronts = 2 per SIMD (64 additions);
  * the same level without the marks (what the marks cost);
  * the one-lane complete addition under the same conditions;
  * dependent launches: an empty kernel, k_pair_level_wide over 1024 ... 32768 additions.
Output: text on stdout (committed as profiles/r6_wide_level_breakdown.txt)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gpu-acceleration_amd"))
from mopro_msm_hip import testhooks  # noqa: E402

MARKS = ["", "operands loaded (LDS)", "stage-1 product (fp_mul)", "exchange + P / R (2 shuffles, 2 fp_sub)", "squares (fp_sqr)",
         "special-case vote (ballot)", "exchange + stage-3 operands (2 shuffles, selects)", "stage-3 product (fp_mul)",
         "exchange + X3 + stage-4 operands (2 shuffles, dbl, add, 2 sub)", "stage-4 product (fp_mul)", "exchange + Y3 (1 shuffle, fp_sub)",
         "result stored (LDS)", "loop left", "barrier passed"]


def main():
    iters = 400
    with testhooks.HooksContext(device=0) as ctx:
        # warm the clock: the probes are single-workgroup kernels, the device would sit at its idle clock otherwise
        for _ in range(3):
            ctx.calibrate()
        print("# one pairwise level e[i] += e[i + m/2] over m XYZZ records in LDS, one workgroup; %d levels per run" % iters)
        for threads, m in ((64, 16), (256, 64), (512, 128), (512, 16)):
            ctx.calibrate()
            plain = ctx.probe_wide_level(threads, m, iters, 0)
            ctx.calibrate()
            marked = ctx.probe_wide_level(threads, m, iters, 1)
            ctx.calibrate()
            scalar = ctx.probe_wide_level(threads, m, iters, 2) if m // 2 <= threads else None
            ghz = plain[0] / (plain[1] * 10.0) if plain[1] else 0.0  # cycles per ns: the constant counter ticks every 10 ns
            print(f"\n== {threads} threads ({threads // 64} wavefronts), {m // 2} additions per level; shader clock {ghz:.3f} GHz")
            print(f"   eight lanes per addition, unmarked: {plain[0] / iters:9.0f} cycles = {plain[1] * 0.01 / iters:6.3f} us per level")
            print(f"   eight lanes per addition, marked:   {marked[0] / iters:9.0f} cycles = {marked[1] * 0.01 / iters:6.3f} us per level")
            if scalar:
                print(f"   one lane per addition (xyzz_add):   {scalar[0] / iters:9.0f} cycles = {scalar[1] * 0.01 / iters:6.3f} us per level")
            tot = sum(marked[2 + k] for k in range(1, 14))
            for k in range(1, 14):
                cyc = marked[2 + k] / iters
                print(f"     {k:2d} {MARKS[k]:62s} {cyc:8.0f} cycles {cyc / (ghz * 1e3) if ghz else 0:6.3f} us {100.0 * marked[2 + k] / tot:5.1f} %")
        print("\n# dependent launches back to back on one stream (hipEvents around 200 launches)")
        for n_adds in (0, 8, 1024, 4096, 8192, 16384, 32768):
            ctx.calibrate()
            us = ctx.probe_launch_chain(n_adds, 200)
            what = "empty kernel" if n_adds == 0 else f"k_pair_level_wide, {n_adds} additions ({n_adds * 8 // 64} wavefronts)"
            print(f"   {what:60s} {us:7.2f} us per launch")
        print("\n# dependent launches of an EMPTY kernel whose workgroups leave at once: blocks x threads, static LDS per workgroup")
        for blocks, threads, lds in ((1, 64, 0), (256, 256, 0), (1152, 512, 0), (1152, 512, 1), (1152, 512, 36), (288, 512, 36), (136, 512, 36), (1152, 64, 0), (8, 512, 36)):
            ctx.calibrate()
            us = ctx.probe_empty_launch(blocks, threads, lds)
            print(f"   {blocks:5d} x {threads:4d} threads ({blocks * threads // 64:5d} wavefronts), {lds:2d} KB LDS {us:7.2f} us per launch")
        print("\n# k_reduce_bits_wide, 2^20 shape (128 workgroups of 512 threads), parts switched off; dependent launches back to back")
        for parts, what in ((7, "whole kernel"), (3, "without the final XYZZ -> Jacobian -> R = 2^256 conversion"), (5, "without the LDS tree"),
                            (6, "without the staging (HBM -> LDS)"), (1, "staging only"), (2, "tree only"), (4, "final conversion only"), (0, "nothing (launch + flag copy)")):
            ctx.calibrate()
            print(f"   {what:70s} {ctx.probe_reduce_bits(parts):7.2f} us per launch")


if __name__ == "__main__":
    main()
