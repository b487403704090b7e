"""Throughput of back-to-back MSMs on ONE GPU: (a) inputs in HBM, one context vs two/three contexts driven by as many host threads
(the other context's sort, bucket reduction and host finish run beside the first one's accumulation) -- the measurement behind
msm_bn254_g1_resident_batch; (b) the product entry itself: resident bases + host scalar vectors, single calls vs the batch call."""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gpu-acceleration_amd")]
import numpy as np
import torch
import mopro_msm_hip as mh
from mopro_msm_hip import testhooks as th
gen = th.HooksContext()
SIZES = [int(x) for x in sys.argv[1].split(',')] if len(sys.argv) > 1 else [14, 17, 20, 22]
DEVICE_PART = len(sys.argv) <= 2
for lg in SIZES:
    n = 1 << lg
    d_b = torch.empty(n * 16, dtype=torch.int32, device="cuda:0"); d_s = torch.empty(n * 8, dtype=torch.int32, device="cuda:0")
    gen.generate_device(1, 2, n, d_b.data_ptr(), d_s.data_ptr()); torch.cuda.synchronize()
    K = 200 if lg <= 20 else 60
    for nctx in ((1, 2, 3) if DEVICE_PART else ()):
        ctxs = [mh.MsmContext() for _ in range(nctx)]
        for c in ctxs:
            for _ in range(3): ref = c.msm_device(d_b.data_ptr(), d_s.data_ptr(), n).jacobian_mont.copy()
        outs = [None] * nctx
        def work(i):
            c = ctxs[i]
            for _ in range(K): r = c.msm_device(d_b.data_ptr(), d_s.data_ptr(), n)
            outs[i] = r.jacobian_mont.copy()
        ths = [threading.Thread(target=work, args=(i,)) for i in range(nctx)]
        t0 = time.perf_counter()
        for t in ths: t.start()
        for t in ths: t.join()
        dt = time.perf_counter() - t0
        ok = all((mh.combine_partials(o.reshape(1, 24)).affine_std == mh.combine_partials(ref.reshape(1, 24)).affine_std).all() for o in outs)
        print(f"2^{lg} device inputs, contexts {nctx}: {nctx * K / dt:8.1f} MSM/s = {dt * 1e3 / (nctx * K):.4f} ms per MSM  same={ok}", flush=True)
        for c in ctxs: c.close()
    # the product entry: resident bases, host scalars (pinned and pageable)
    c = mh.MsmContext()
    c.upload_bases(d_b.cpu().numpy().view(np.uint32).reshape(n, 16), mh.FORM_MONT)
    hs_t = d_s.cpu()
    for name, t in (("pageable", hs_t), ("pinned", hs_t.pin_memory())):
        hs = t.numpy().view(np.uint32).reshape(n, 8)
        B = 16
        ref = c.msm_resident(hs).affine_std
        for _ in range(2): c.msm_resident_batch([hs] * 4)
        t0 = time.perf_counter()
        for _ in range(B): r1 = c.msm_resident(hs)
        t1 = time.perf_counter()
        rb = c.msm_resident_batch([hs] * B)
        t2 = time.perf_counter()
        ok = all((x.affine_std == ref).all() for x in rb) and (r1.affine_std == ref).all()
        print(f"2^{lg} resident bases, {name} scalars: single calls {(t1 - t0) * 1e3 / B:.4f} ms per MSM, batch of {B} "
              f"{(t2 - t1) * 1e3 / B:.4f} ms per MSM  same={ok}", flush=True)
    c.close()
gen.close()
