"""N = 2^22 host-pointer path: pinned vs pageable caller memory, streamed vs single-shot, hipHostRegister cost (GPU box)."""
import os, sys, time
ROOT = "/root/repo" if os.path.exists("/root/repo/bench.py") else os.getcwd()
sys.path[:0] = [ROOT, os.path.join(ROOT, "gpu-acceleration_amd")]
import numpy as np, torch
import mopro_msm_hip as mh
from mopro_msm_hip import testhooks as th
GEN = th.HooksContext()  # the synthetic-instance generator lives in the hooks build


def main():
    n = 1 << 22
    with mh.MsmContext(stream_chunk_log2=28) as c0:
        d_b = torch.empty(n * 16, dtype=torch.int32, device="cuda:0"); d_s = torch.empty(n * 8, dtype=torch.int32, device="cuda:0")
        GEN.generate_device(1, 2, n, d_b.data_ptr(), d_s.data_ptr())
        tb = d_b.cpu(); ts = d_s.cpu()
        t=time.perf_counter(); pb = tb.pin_memory(); ps = ts.pin_memory(); print("pin_memory copy ms", (time.perf_counter()-t)*1e3)
        hb = pb.numpy().view(np.uint32).reshape(n, 16); hs = ps.numpy().view(np.uint32).reshape(n, 8)
        for _ in range(2): r0 = c0.msm(hb, hs, mh.FORM_MONT)
        t = time.perf_counter(); r0 = c0.msm(hb, hs, mh.FORM_MONT); print("pinned single-shot", (time.perf_counter() - t) * 1e3, c0.timings())
    for lg in (19, 20):
        with mh.MsmContext(stream_chunk_log2=lg) as c1:
            for _ in range(2): r1 = c1.msm(hb, hs, mh.FORM_MONT)
            t = time.perf_counter(); r1 = c1.msm(hb, hs, mh.FORM_MONT); print("pinned streamed", lg, (time.perf_counter() - t) * 1e3)
    # hipHostRegister cost
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    a = np.zeros(n*16, np.uint32)
    t=time.perf_counter(); rc = hip.hipHostRegister(ctypes.c_void_p(a.ctypes.data), ctypes.c_size_t(a.nbytes), 0); print("hipHostRegister 256MB rc",rc,"ms",(time.perf_counter()-t)*1e3)
    t=time.perf_counter(); rc = hip.hipHostUnregister(ctypes.c_void_p(a.ctypes.data)); print("unregister ms",(time.perf_counter()-t)*1e3)



if __name__ == "__main__":
    main()
