"""Timeline of ONE host-pointer call (pinned, 2^20 by default): kernels AND host->device copies, in start order.
usage: rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d DIR -- python3 tools/host_call_timeline.py [logn] ;
       python3 tools/host_call_timeline.py --summarise DIR"""
import csv, glob, os, sys
if len(sys.argv) > 2 and sys.argv[1] == "--summarise":
    ev = []
    for f in glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "").replace("msmk::", "")))
    for f in glob.glob(sys.argv[2] + "/**/*memory_copy_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY %s %s B" % (r.get("Direction", "?"), r.get("Bytes", r.get("Size", "?")))))
    ev.sort()
    ends = [i for i, e in enumerate(ev) if e[2].startswith("k_reduce_bits")]
    i0, i1 = ends[-3] + 1, ends[-2] + 1  # the second to last call
    t0 = ev[i0][0]
    for s, e, name in ev[i0:i1]:
        print(f"{(s - t0) / 1e3:9.1f} .. {(e - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  {name}")
    sys.exit(0)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gpu-acceleration_amd")]
import time, numpy as np, torch
import mopro_msm_hip as mh
from mopro_msm_hip import testhooks as th
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << lg
gen = th.HooksContext()
d_b = torch.empty(n * 16, dtype=torch.int32, device="cuda:0"); d_s = torch.empty(n * 8, dtype=torch.int32, device="cuda:0")
gen.generate_device(1, 2, n, d_b.data_ptr(), d_s.data_ptr()); torch.cuda.synchronize()
tb, ts = d_b.cpu().pin_memory(), d_s.cpu().pin_memory()
hb, hs = tb.numpy().view(np.uint32).reshape(n, 16), ts.numpy().view(np.uint32).reshape(n, 8)
with mh.MsmContext() as c:
    t = []
    for _ in range(12):
        t0 = time.perf_counter(); c.msm(hb, hs, mh.FORM_MONT); t.append((time.perf_counter() - t0) * 1e3)
    print("ms per call:", " ".join(f"{x:.3f}" for x in t))
