#!/usr/bin/env python3
"""Round 6 (NOTES_r6 section 15): the wider net behind tools/race_hunt.py -- MEDIUM sizes (4097 .. 2^16 points: single-shot host calls whose bases travel on the copy
stream beside the sort), struct arrays with and without points at infinity (the optimistic pass and its repeat), device calls, the resident set and its batch call,
through kept and fresh contexts in random order; expected points in closed form ((sum s_i k_i) G for bases k_i G).  Prints the mismatch count.
usage: tools/race_hunt_wide.py [calls] [seed]     (hooks build as engine: MSM_HIP_NO_POLL / MSM_HIP_POLL_VERIFY apply)"""
import os, sys, time
os.environ.setdefault("MSM_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpu-acceleration_amd", "libmsm_hip_hooks.so"))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gpu-acceleration_amd")]
import numpy as np
import torch
import mopro_msm_hip as mh
from mopro_msm_hip import testhooks as th
from oracle import bn254_oracle as orc

want = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
N = 1 << 16
k = th.generate_scalars_host(0xB2540D01, N, nonzero=True)
s = th.generate_scalars_host(0xB2540D02, N)
bases = orc.gen_bases_from_logs(k, orc.FORM_MONT)
rinv = pow(1 << 256, -1, orc.R_ORDER)
g = np.zeros(16, np.uint32)
g[0], g[8] = 1, 2
expect = lambda d: orc.g1_to_affine_std(orc.g1_scalar_mul(g, orc.int_to_words(d % orc.R_ORDER)))[0]
img = np.zeros((N, 72), np.uint8)
img[:, :64] = bases.view(np.uint8).reshape(N, 64)
SIZES = [4097, 8192, 1 << 14, 1 << 16]
inf_idx = {m: np.array([0, 3, m // 2, m - 1]) for m in SIZES}
E = {}
for m in SIZES:
    d = orc.dot_words(k[:m], s[:m])
    sm = s[:m].copy()
    sm[inf_idx[m]] = 0
    E[m] = dict(std=expect(d), mont=expect(d * rinv), mont_inf=expect(orc.dot_words(k[:m], sm) * rinv))
img_inf = {}
for m in SIZES:
    im = img[:m].copy()
    im[inf_idx[m], 64] = 1
    im[inf_idx[m], :64] = 0
    img_inf[m] = im
d_b = torch.from_numpy(bases.view(np.int32).reshape(-1).copy()).cuda()
d_s = torch.from_numpy(s.view(np.int32).reshape(-1).copy()).cuda()
cfgs = [(0, 0), (0, mh.FLAG_NO_GLV), (13, mh.FLAG_NO_GLV), (0, mh.FLAG_WINDOW_TABLE)]
kept = {c: mh.MsmContext(window_bits=c[0], flags=c[1]) for c in cfgs}
bad, calls = {}, 0
t0 = time.time()
def check(tag, r, exp):
    global calls
    calls += 1
    if r.is_infinity or not (r.affine_std == exp).all():
        bad[tag] = bad.get(tag, 0) + 1
        print("MISMATCH", tag, "call", calls, flush=True)
while calls < want:
    m = SIZES[int(rng.integers(0, len(SIZES) - (0 if rng.random() < 0.15 else 1)))]  # (2^16 less often)
    cfg = cfgs[int(rng.integers(0, len(cfgs)))]
    fresh = rng.random() < 0.1
    ctx = mh.MsmContext(window_bits=cfg[0], flags=cfg[1]) if fresh else kept[cfg]
    tag = (m, cfg, "fresh" if fresh else "kept")
    kind = int(rng.integers(0, 6))
    if kind == 0:
        check(tag + ("host words",), ctx.msm(bases[:m], s[:m], mh.FORM_MONT), E[m]["std"])
    elif kind == 1:
        check(tag + ("structs",), ctx.msm_arkworks(img[:m], 72, 0, 32, 64, s[:m]), E[m]["mont"])
    elif kind == 2:
        check(tag + ("structs with infinity",), ctx.msm_arkworks(img_inf[m], 72, 0, 32, 64, s[:m]), E[m]["mont_inf"])
        check(tag + ("structs after infinity",), ctx.msm_arkworks(img[:m], 72, 0, 32, 64, s[:m]), E[m]["mont"])
    elif kind == 3:
        check(tag + ("device",), ctx.msm_device(d_b.data_ptr(), d_s.data_ptr(), m), E[m]["std"])
        check(tag + ("device again",), ctx.msm_device(d_b.data_ptr(), d_s.data_ptr(), m), E[m]["std"])
    elif kind == 4:
        ctx.upload_bases(bases[:m], mh.FORM_MONT)
        check(tag + ("resident",), ctx.msm_resident(s[:m]), E[m]["std"])
        check(tag + ("host after resident",), ctx.msm(bases[:m], s[:m], mh.FORM_MONT), E[m]["std"])
    else:
        ctx.upload_bases(bases[:m], mh.FORM_MONT)
        for r in ctx.msm_resident_batch([s[:m]] * 4):
            check(tag + ("resident batch",), r, E[m]["std"])
    if fresh:
        ctx.close()
print(f"race_hunt_wide: {calls} calls, {sum(bad.values())} mismatches, {time.time() - t0:.1f} s, NO_POLL={os.environ.get('MSM_HIP_NO_POLL')}")
for kk, v in bad.items():
    print("  ", kk, v)
