"""A few resident 2^N MSMs (msm_bn254_g1_device) for rocprofv3 --kernel-trace: the per-kernel timeline of ONE call, gaps included.
usage: rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/device_call_trace.py 20 ; then summarise with
python3 tools/device_call_trace.py --summarise DIR"""
import csv, glob, os, sys
if len(sys.argv) > 2 and sys.argv[1] == "--summarise":
    ev = []
    for f in glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "").replace("msmk::", "")))
    ev.sort()
    # the last call = from the last k_decompose* on
    starts = [i for i, e in enumerate(ev) if e[2].startswith("k_decompose")]
    i0 = starts[-2]  # second to last call (the last one may be cut)
    i1 = starts[-1]
    t0 = ev[i0][0]
    prev_end = t0
    tot_k = 0
    for s, e, name in ev[i0:i1]:
        print(f"{(s - t0) / 1e3:9.1f} us  gap {(s - prev_end) / 1e3:6.1f}  dur {(e - s) / 1e3:8.1f}  {name}")
        tot_k += e - s
        prev_end = max(prev_end, e)
    print(f"kernels {tot_k / 1e3:.1f} us, span {(prev_end - t0) / 1e3:.1f} us, next call starts {(ev[i1][0] - prev_end) / 1e3:.1f} us after the last kernel")
    sys.exit(0)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gpu-acceleration_amd")]
import time, torch
import mopro_msm_hip as mh
from mopro_msm_hip import testhooks as th
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << lg
gen = th.HooksContext()
d_b = torch.empty(n * 16, dtype=torch.int32, device="cuda:0"); d_s = torch.empty(n * 8, dtype=torch.int32, device="cuda:0")
gen.generate_device(1, 2, n, d_b.data_ptr(), d_s.data_ptr()); torch.cuda.synchronize()
c = mh.MsmContext(max_points=n, window_bits=int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.2: c.msm_device(d_b.data_ptr(), d_s.data_ptr(), n)
ts = []
for _ in range(10):
    t = time.perf_counter(); c.msm_device(d_b.data_ptr(), d_s.data_ptr(), n); ts.append((time.perf_counter() - t) * 1e3)
print("ms per call:", " ".join(f"{x:.3f}" for x in ts))
