"""Soak of msm_bn254_g1_resident_batch: random batch sizes and truncations on resident sets of several sizes (both batch modes), every
result against the single resident call; interleaved with other entry points of the same context."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gpu-acceleration_amd")]
import numpy as np, torch
import mopro_msm_hip as mh
from mopro_msm_hip import testhooks as th
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 30
TABLE = len(sys.argv) > 2 and sys.argv[2] == "table"  # the resident sets carry their window table (MSM_FLAG_WINDOW_TABLE); checked against a plain context
rng = np.random.default_rng(20261003)
gen = th.HooksContext()
nmax = 1 << 20
d_b = torch.empty(nmax * 16, dtype=torch.int32, device="cuda:0"); d_s = torch.empty(nmax * 8, dtype=torch.int32, device="cuda:0")
gen.generate_device(5, 6, nmax, d_b.data_ptr(), d_s.data_ptr()); torch.cuda.synchronize()
hb = d_b.cpu().numpy().view(np.uint32).reshape(nmax, 16)
pool = [th.generate_scalars_host(100 + j, nmax) for j in range(4)]
c = mh.MsmContext(flags=mh.FLAG_WINDOW_TABLE if TABLE else 0)
ref = mh.MsmContext() if TABLE else c
t0 = time.time(); calls = msms = bad = 0
while time.time() - t0 < secs:
    lg = int(rng.choice([10, 14, 17, 18, 19, 20]))
    n = (1 << lg) - int(rng.integers(0, 3)) * 7
    c.upload_bases(hb[:n], mh.FORM_MONT)
    if TABLE: ref.upload_bases(hb[:n], mh.FORM_MONT)
    for _ in range(int(rng.integers(1, 4))):
        k = int(rng.integers(1, 7))
        m = n if rng.random() < 0.7 else int(rng.integers(1, n + 1))
        vecs = [pool[int(rng.integers(0, 4))][int(o):int(o) + m] for o in rng.integers(0, nmax - m + 1, size=k)]
        res = c.msm_resident_batch(vecs)
        for v, r in zip(vecs, res):
            one = ref.msm_resident(v)
            if one.is_infinity != r.is_infinity or not (one.jacobian_mont == r.jacobian_mont).all():
                if not (one.affine_std == r.affine_std).all(): bad += 1
        calls += 1; msms += k
        if rng.random() < 0.3: c.msm_device(d_b.data_ptr(), d_s.data_ptr(), 1 << int(rng.integers(8, 18)))
print(f"batch_soak{' (window table)' if TABLE else ''}: {calls} batch calls, {msms} MSMs, mismatches {bad}, {time.time() - t0:.1f} s")
