#!/usr/bin/env python3
"""Host cost of msm_bn254_g1_combine (fold of G partials + canonical affine) -- no GPU needed."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gpu-acceleration_amd")]
import numpy as np
import mopro_msm_hip as mh
from oracle import bn254_oracle as orc
g = np.zeros(16, np.uint32); g[0], g[8] = 1, 2
parts = np.stack([orc.g1_scalar_mul(g, orc.int_to_words(1234567 + 99991 * i)) for i in range(8)])
same = np.repeat(parts[:1], 8, axis=0)
for name, arr in (("8 distinct partials", parts), ("8 equal partials", same)):
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(300): mh.combine_partials(arr)
        print(name, "rep", rep, f"{(time.perf_counter() - t0) / 300 * 1e6:.1f} us per fold")
