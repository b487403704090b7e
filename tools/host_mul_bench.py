#!/usr/bin/env python3
"""Host field arithmetic speed: ns per Jacobian addition (16 multiplications) through msm_bn254_g1_combine.  No GPU used."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gpu-acceleration_amd")]
import numpy as np, mopro_msm_hip as mh
from oracle import bn254_oracle as orc
g = np.zeros(16, np.uint32); g[0], g[8] = 1, 2
parts = np.stack([orc.g1_scalar_mul(g, orc.int_to_words(1234567 + 99991 * i)) for i in range(64)])
big = np.tile(parts, (256, 1))  # 16384 partials
exp = mh.combine_partials(big).affine_std
best = 1e9
for rep in range(7):
    t0 = time.perf_counter(); r = mh.combine_partials(big, want_affine=False); dt = time.perf_counter() - t0
    best = min(best, dt / big.shape[0] * 1e9)
assert (r.affine_std == exp).all()
print(f"{os.environ.get('MSM_HIP_LIB', 'in-tree lib')}: {best:.0f} ns per jadd (best of 7), {best/16:.1f} ns per field multiplication")
