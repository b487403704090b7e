#!/usr/bin/env python3
"""Randomised parity run on the GPU box: random sizes, window sizes, digit modes, infinity masks and scalar skews,
every result compared bit for bit with the CPU oracle (Pippenger) -- a wider net than the fixed test cases.
usage: tools/fuzz_parity.py [cases] [seed] [only_case [repeats]]   (only_case: replay the random draws, run THAT case `repeats` times and say which check fails)"""
import os, sys, time
os.environ.setdefault("MSM_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpu-acceleration_amd", "libmsm_hip_hooks.so"))  # the A/B knobs this script sets are read by the HOOKS build only (round 5)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gpu-acceleration_amd")]
import numpy as np
import mopro_msm_hip as mh
from oracle import bn254_oracle as orc

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 20261002)
ONLY = int(sys.argv[3]) if len(sys.argv) > 3 else None
REPEATS = int(sys.argv[4]) if len(sys.argv) > 4 else 1
NMAX = 60000
k_all = orc.gen_scalars(4242, NMAX, nonzero=True)
bases_all = orc.gen_bases_from_logs(k_all, orc.FORM_MONT)
s_all = orc.gen_scalars(4343, NMAX)
t0 = time.time()
bad = 0
for it in range(cases):
    n = int(rng.choice([rng.integers(1, 40), rng.integers(40, 3000), rng.integers(3000, NMAX)]))
    off = int(rng.integers(0, NMAX - n + 1))
    bases = bases_all[off:off + n].copy()
    mode = int(rng.integers(0, 8))
    s = s_all[off:off + n].copy()
    if mode == 1: s[:] = s[0]                                   # all equal
    elif mode == 2: s = s[np.arange(n) % 3]                     # 3 distinct
    elif mode == 3: s[:, 1:] = 0                                # < 2^32
    elif mode == 4: s[rng.random(n) < 0.6] = 0                  # many zeros
    elif mode == 5:                                             # witness-like
        u = rng.random(n); s[u < 0.7] = 0; s[(u >= 0.3) & (u < 0.7), 0] = 1
    elif mode == 6: s[:] = orc.int_to_words(orc.R_ORDER - 1)    # all r-1
    elif mode == 7 and n > 1: bases[1::2] = bases[0]            # duplicate bases
    inf = None
    if rng.random() < 0.4:
        inf = (rng.random(n) < rng.choice([0.001, 0.05, 0.9])).astype(np.uint8)
    wb = int(rng.choice([0, 0, 0, 2, 3, 5, 8, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20]))
    flags = mh.FLAG_UNSIGNED_DIGITS if (rng.random() < 0.25 and wb not in (17, 18, 19, 20)) else 0
    if rng.random() < 0.35:
        flags |= mh.FLAG_NO_GLV
    table = rng.random() < 0.4  # row f4: the resident path with its window table (shared bucket arrays), random factor
    for kk in ("MSM_HIP_TABLE_F",):
        os.environ.pop(kk, None)
    if table:
        flags |= mh.FLAG_WINDOW_TABLE
        pl = mh.plan(n, wb, flags)
        divs = [f for f in range(2, pl.num_windows + 1) if pl.num_windows % f == 0]
        if divs and rng.random() < 0.5:
            os.environ["MSM_HIP_TABLE_F"] = str(int(rng.choice(divs)))
    if ONLY is not None and it != ONLY:
        continue
    for rep in range(REPEATS if ONLY is not None else 1):
        with mh.MsmContext(window_bits=wb, flags=flags) as ctx:
            if table:
                ctx.upload_bases(bases, mh.FORM_MONT, inf)
                r = ctx.msm_resident(s)
                r2 = ctx.msm_resident_batch([s, s])[1]
            else:
                r = ctx.msm(bases, s, mh.FORM_MONT, inf)
                r2 = ctx.msm(bases, s, mh.FORM_MONT, inf)
        exp, einf, _ = orc.msm_pippenger(bases, s, orc.FORM_MONT, inf)
        ok1 = r.is_infinity == bool(einf) and (r.affine_std == exp).all()
        ok2 = r2.is_infinity == bool(einf) and (r2.affine_std == exp).all()
        if not (ok1 and ok2):
            bad += 1
            print("MISMATCH case", it, "rep", rep, "first call ok", bool(ok1), "second call ok", bool(ok2),
                  dict(n=n, off=off, mode=mode, wb=wb, flags=flags, table_f=os.environ.get('MSM_HIP_TABLE_F'), inf=None if inf is None else int(inf.sum())), flush=True)
print(f"fuzz_parity: {cases} cases, {bad} mismatches, {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
