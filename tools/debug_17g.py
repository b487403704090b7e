#!/usr/bin/env python3
"""debug: c = 17 GLV window table at 2^21 returned a wrong point (tools/table_sweep.py 17g).  Variations to localise it."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gpu-acceleration_amd")]
import numpy as np
import torch
import mopro_msm_hip as mh
from mopro_msm_hip import testhooks as th
from oracle import bn254_oracle as orc

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 21
n = 1 << logn
seed = 0xB2540300 + logn
d_b = torch.empty(n * 16, dtype=torch.int32, device="cuda:0")
d_s = torch.empty(n * 8, dtype=torch.int32, device="cuda:0")
with th.HooksContext() as gen:
    gen.generate_device(seed, seed + 1, n, d_b.data_ptr(), d_s.data_ptr())
torch.cuda.synchronize()
hb = d_b.cpu().numpy().view(np.uint32).reshape(n, 16)
v = th.generate_scalars_host(seed + 1, n)
k = th.generate_scalars_host(seed, n, nonzero=True)
exp = orc.closed_form_expected(k, v)[0]


def run(name, env, wb=0, flags=0, resident=True):
    for kk in ("MSM_HIP_TABLE_C", "MSM_HIP_TABLE_F", "MSM_HIP_TABLE_GLV_MAX_LOG2", "MSM_HIP_CHUNK_LEN", "MSM_HIP_GLV_MAX_LOG2"):
        os.environ.pop(kk, None)
    os.environ.update(env)
    with mh.MsmContext(window_bits=wb, flags=flags) as c:
        if resident:
            c.upload_bases(hb, mh.FORM_MONT)
            r = c.msm_resident(v)
            r2 = c.msm_resident(v)
        else:
            r = c.msm_device(d_b.data_ptr(), d_s.data_ptr(), n)
            r2 = c.msm_device(d_b.data_ptr(), d_s.data_ptr(), n)
        pl = mh.plan(n, wb, flags)
        print("%-44s c %d W %d f %d glv %d: exact %s / %s" % (name, pl.window_bits, pl.num_windows, pl.table_factor, pl.glv,
              bool((r.affine_std == exp).all()), bool((r2.affine_std == exp).all())), flush=True)


T = mh.FLAG_WINDOW_TABLE
run("17g table", {"MSM_HIP_TABLE_C": "17", "MSM_HIP_TABLE_GLV_MAX_LOG2": "23"}, flags=T)
run("17g table, chunk 32", {"MSM_HIP_TABLE_C": "17", "MSM_HIP_TABLE_GLV_MAX_LOG2": "23", "MSM_HIP_CHUNK_LEN": "32"}, flags=T)
run("17g table, chunk 512", {"MSM_HIP_TABLE_C": "17", "MSM_HIP_TABLE_GLV_MAX_LOG2": "23", "MSM_HIP_CHUNK_LEN": "512"}, flags=T)
run("c17 GLV, no table (device call)", {"MSM_HIP_GLV_MAX_LOG2": "23"}, wb=17, resident=False)
run("c17 unsplit table", {"MSM_HIP_TABLE_C": "17"}, flags=T)
run("c18 GLV table", {"MSM_HIP_TABLE_C": "18", "MSM_HIP_TABLE_GLV_MAX_LOG2": "23"}, flags=T)
run("c15 GLV table", {"MSM_HIP_TABLE_C": "15", "MSM_HIP_TABLE_GLV_MAX_LOG2": "23"}, flags=T)
run("17g table f=4", {"MSM_HIP_TABLE_C": "17", "MSM_HIP_TABLE_GLV_MAX_LOG2": "23", "MSM_HIP_TABLE_F": "4"}, flags=T)
