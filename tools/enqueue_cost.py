"""How much of a resident call is host-side enqueueing (what a hipGraph could shorten)?  MSM_HIP_TRACE=1 prints `host enqueue` per call;
note that tracing turns the per-stage events on (each ~6 us of stream time), so totals here sit above the untraced latency."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, re
sys.path[:0] = [%r, %r]
import torch, mopro_msm_hip as mh
from mopro_msm_hip import testhooks as th
gen = th.HooksContext()
for lg in (10, 14, 16, 17, 18, 20):
    n = 1 << lg
    d_b = torch.empty(n * 16, dtype=torch.int32, device="cuda:0"); d_s = torch.empty(n * 8, dtype=torch.int32, device="cuda:0")
    gen.generate_device(1, 2, n, d_b.data_ptr(), d_s.data_ptr()); torch.cuda.synchronize()
    with mh.MsmContext() as c:
        for _ in range(8): c.msm_device(d_b.data_ptr(), d_s.data_ptr(), n)
''' % (ROOT, os.path.join(ROOT, "gpu-acceleration_amd"))
p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, MSM_HIP_TRACE="1"))
import re, statistics, collections
rows = collections.defaultdict(list)
for l in p.stderr.splitlines():
    m = re.search(r"\] device dev \d+ n (\d+) .* total ([0-9.]+) ms \(host enqueue ([0-9.]+) ms\)", l)
    if m: rows[int(m.group(1))].append((float(m.group(2)), float(m.group(3))))
for n, v in sorted(rows.items()):
    v = v[3:]
    print(f"n = 2^{n.bit_length()-1}: total (traced) {statistics.median(x[0] for x in v):.3f} ms, host enqueue {statistics.median(x[1] for x in v):.3f} ms")
if not rows: print(p.stderr[-2000:])
