#!/usr/bin/env python3
"""Host-pointer call (pinned memory, arkworks-form words) over product-library builds, whole processes interleaved on one box:
    tools/host_path_ab.py [--rounds R] name1 name2 ... [-- log_n ...]      (name = tools/_ab/libmsm_hip_<name>.so, `base` = the in-tree product)
Per build and size: the automatic chunking and explicit 2^17-point chunks (msm_config_t.stream_chunk_log2), median of the rounds' medians."""
import json, os, statistics, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, time, json
sys.path[:0] = [%r, os.path.join(%r, "gpu-acceleration_amd")]
import numpy as np, torch
import mopro_msm_hip as mh
from mopro_msm_hip import testhooks as th
out = {}
for lg in [int(a) for a in sys.argv[1:]]:
    n = 1 << lg
    d_b = torch.empty(n * 16, dtype=torch.int32, device="cuda:0"); d_s = torch.empty(n * 8, dtype=torch.int32, device="cuda:0")
    with th.HooksContext() as gen:
        gen.generate_device(1, 2, n, d_b.data_ptr(), d_s.data_ptr())
    torch.cuda.synchronize()
    pb, ps = d_b.cpu().pin_memory(), d_s.cpu().pin_memory()
    hb, hs = pb.numpy().view(np.uint32).reshape(n, 16), ps.numpy().view(np.uint32).reshape(n, 8)
    for label, chunk in (("auto", 0), ("chunk17", 17)):
        with mh.MsmContext(stream_chunk_log2=chunk) as c:
            t_w = time.perf_counter()
            while time.perf_counter() - t_w < 0.15:
                c.msm(hb, hs, mh.FORM_MONT)
            ts = []
            for _ in range(15):
                t = time.perf_counter(); c.msm(hb, hs, mh.FORM_MONT); ts.append((time.perf_counter() - t) * 1e3)
            ts.sort()
            out["2^%%d %%s" %% (lg, label)] = [ts[len(ts) // 2], c.timings()["stream_chunks"]]
print(json.dumps(out))
''' % (ROOT, ROOT)

args = sys.argv[1:]
sizes = ["19", "20"]
if "--" in args:
    k = args.index("--"); sizes = args[k + 1:]; args = args[:k]
rounds = 3
if args and args[0] == "--rounds":
    rounds = int(args[1]); args = args[2:]
res = {v: {} for v in args}
for rnd in range(rounds):
    for v in (args if rnd % 2 == 0 else args[::-1]):
        env = dict(os.environ)
        env.pop("MSM_HIP_LIB", None)
        if v != "base":
            env["MSM_HIP_LIB"] = os.path.join(ROOT, "tools", "_ab", "libmsm_hip_%s.so" % v)
        p = subprocess.run([sys.executable, "-c", WORKER] + sizes, capture_output=True, text=True, env=env)
        try:
            j = json.loads(p.stdout.strip().splitlines()[-1])
        except Exception:
            print(v, "FAILED", p.stdout[-300:], p.stderr[-800:], flush=True)
            continue
        for k, (ms, ch) in j.items():
            res[v].setdefault(k, []).append((ms, ch))
for v in args:
    for k, xs in res[v].items():
        print("%-8s %-14s median %.4f ms  (%s)  chunks %d" % (v, k, statistics.median(x[0] for x in xs), ", ".join("%.3f" % x[0] for x in xs), xs[0][1]), flush=True)
