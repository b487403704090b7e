#!/usr/bin/env python3
"""Round 6: the 24 MSM known answers the reference ships (tests/golden/srs_kzg_points.json) through the host call and the resident set, `reps` fresh contexts per
plan (window table / split / unsplit) -- a repeat count the gate cannot afford: python3 tools/stress_known_answers.py 40"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gpu-acceleration_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import mopro_msm_hip as mh
from conftest import load_srs_sets
from oracle import bn254_oracle as orc
R = orc.R_ORDER
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
bad = 0
tot = 0
sets = load_srs_sets()
pre = {}
for fname, k, omega, g, gl in sets:
    n = 1 << k
    for j in range(n):
        pre[(fname, j)] = (np.stack([orc.int_to_words(pow(omega, i * j, R)) for i in range(n)]), np.concatenate([orc.fq_from_mont(g[j, :8]), orc.fq_from_mont(g[j, 8:])]))
for wb, flags in [(0, mh.FLAG_WINDOW_TABLE), (0, 0), (0, mh.FLAG_NO_GLV)]:
    for rep in range(reps):
        for fname, k, omega, g, gl in sets:
            n = 1 << k
            with mh.MsmContext(window_bits=wb, flags=flags) as c:
                c.upload_bases(gl, mh.FORM_MONT)
                for j in range(n):
                    scalars, exp = pre[(fname, j)]
                    r = c.msm(gl, scalars, mh.FORM_MONT)
                    ok1 = (not r.is_infinity) and (r.affine_std == exp).all()
                    ok2 = (c.msm_resident(scalars).affine_std == exp).all()
                    tot += 2
                    if not ok1 or not ok2:
                        bad += 1
                        print("MISMATCH flags", flags, "rep", rep, fname, j, "host ok", ok1, "resident ok", ok2, flush=True)
    print("flags", flags, "done; bad so far", bad, "of", tot, flush=True)
