"""k_accumulate chunk length (MSM_HIP_CHUNK_LEN, read per call) vs instance size, one process: median latency of the resident call
and the mean k_accumulate time.  usage: python tools/chunk_len_sweep.py <log_n,...> <L,...>"""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gpu-acceleration_amd")]
import torch
import mopro_msm_hip as mh
from mopro_msm_hip import testhooks as th

sizes = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "16,17,18,20").split(",")]
Ls = (sys.argv[2] if len(sys.argv) > 2 else "0,8,10,11,12,14,16,20,22,24,28,29,30,32").split(",")
gen = th.HooksContext()
for lg in sizes:
    n = 1 << lg
    d_b = torch.empty(n * 16, dtype=torch.int32, device="cuda:0"); d_s = torch.empty(n * 8, dtype=torch.int32, device="cuda:0")
    gen.generate_device(1, 2, n, d_b.data_ptr(), d_s.data_ptr()); torch.cuda.synchronize()
    with mh.MsmContext() as c:
        ref = None
        rows = {L: [] for L in Ls}
        for rnd in range(3):
            for L in Ls:
                if L == "0": os.environ.pop("MSM_HIP_CHUNK_LEN", None)
                else: os.environ["MSM_HIP_CHUNK_LEN"] = L
                for _ in range(3): r = c.msm_device(d_b.data_ptr(), d_s.data_ptr(), n)
                c.reset_kernel_stats()
                ts = []
                for _ in range(25):
                    t = time.perf_counter(); r = c.msm_device(d_b.data_ptr(), d_s.data_ptr(), n); ts.append((time.perf_counter() - t) * 1e3)
                acc, _ = c.accumulate_kernel_stats()
                if ref is None: ref = r.jacobian_mont.copy()
                same = bool((mh.combine_partials(r.jacobian_mont.reshape(1, 24)).affine_std == mh.combine_partials(ref.reshape(1, 24)).affine_std).all())
                rows[L].append((statistics.median(ts), acc, same))
        os.environ.pop("MSM_HIP_CHUNK_LEN", None)
        for L in Ls:
            v = rows[L]
            print(f"2^{lg} L={L:>3s} median {statistics.median(x[0] for x in v):.4f} ms  k_accumulate {statistics.median(x[1] for x in v):.4f} ms  same={all(x[2] for x in v)}", flush=True)
gen.close()
