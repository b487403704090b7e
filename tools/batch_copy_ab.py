"""resident batch, one compute stream: uploads through the owner's copy stream (default) vs each pipeline's own (MSM_HIP_BATCH_COPY=0), same
process, alternating, pinned and pageable scalars."""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gpu-acceleration_amd")]
import numpy as np, torch
import mopro_msm_hip as mh
from mopro_msm_hip import testhooks as th
gen = th.HooksContext()
for lg in [int(a) for a in sys.argv[1:]] or [19, 20, 22]:
    n = 1 << lg
    d_b = torch.empty(n * 16, dtype=torch.int32, device="cuda:0"); d_s = torch.empty(n * 8, dtype=torch.int32, device="cuda:0")
    gen.generate_device(1, 2, n, d_b.data_ptr(), d_s.data_ptr()); torch.cuda.synchronize()
    hs_t = d_s.cpu()
    with mh.MsmContext() as c:
        c.upload_bases(d_b.cpu().numpy().view(np.uint32).reshape(n, 16), mh.FORM_MONT)
        for kind, t in (("pinned", hs_t.pin_memory()), ("pageable", hs_t)):
            hs = t.numpy().view(np.uint32).reshape(n, 8)
            B = 16 if lg <= 20 else 8
            c.msm_resident_batch([hs] * 8)
            res = {"owner": [], "own": []}
            for rnd in range(4):
                for mode in ("owner", "own"):
                    if mode == "own": os.environ["MSM_HIP_BATCH_COPY"] = "0"
                    else: os.environ.pop("MSM_HIP_BATCH_COPY", None)
                    c.msm_resident_batch([hs] * 4)
                    t0 = time.perf_counter(); c.msm_resident_batch([hs] * B); res[mode].append((time.perf_counter() - t0) * 1e3 / B)
            os.environ.pop("MSM_HIP_BATCH_COPY", None)
            print(f"2^{lg} {kind:8s}: owner's copy stream {statistics.median(res['owner']):.4f} ms per MSM {['%.3f' % x for x in res['owner']]}   own {statistics.median(res['own']):.4f} {['%.3f' % x for x in res['own']]}", flush=True)
gen.close()
