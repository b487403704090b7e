"""Host-pointer entry (msm_bn254_g1 / msm_bn254_g1_arkworks) latency vs size, caller memory kind and streaming knobs (GPU box).
usage: python tools/host_path_sweep.py [log_n ...]      env knobs are set per run by this script (fresh context each)."""
import os, sys, time
os.environ.setdefault("MSM_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpu-acceleration_amd", "libmsm_hip_hooks.so"))  # the A/B knobs this script sets are read by the HOOKS build only (round 5)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gpu-acceleration_amd")]
import numpy as np, torch
import mopro_msm_hip as mh
from mopro_msm_hip import testhooks as th


def timed(fn, reps=9):
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < 0.12:  # clock ramp: ~35 ms of work after an idle gap (tools/clock_ramp.py)
        fn()
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); fn(); ts.append((time.perf_counter() - t) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [16, 18, 19, 20, 21, 22]
    gen = th.HooksContext()
    for lg in sizes:
        n = 1 << lg
        d_b = torch.empty(n * 16, dtype=torch.int32, device="cuda:0"); d_s = torch.empty(n * 8, dtype=torch.int32, device="cuda:0")
        gen.generate_device(1, 2, n, d_b.data_ptr(), d_s.data_ptr()); torch.cuda.synchronize()
        tb, ts_ = d_b.cpu(), d_s.cpu()
        pb, ps = tb.pin_memory(), ts_.pin_memory()
        hb, hs = tb.numpy().view(np.uint32).reshape(n, 16), ts_.numpy().view(np.uint32).reshape(n, 8)
        hbp, hsp = pb.numpy().view(np.uint32).reshape(n, 16), ps.numpy().view(np.uint32).reshape(n, 8)
        img = np.zeros((n, 72), np.uint8); img[:, :64] = hb.view(np.uint8).reshape(n, 64)
        with mh.MsmContext() as c:
            med, mn = timed(lambda: c.msm_device(d_b.data_ptr(), d_s.data_ptr(), n))
        print(f"2^{lg} resident                      median {med:7.3f} min {mn:7.3f} ms", flush=True)
        configs = [("auto", {})]
        if lg >= 18:
            configs += [("single-shot", {"MSM_HIP_STREAM_MIN_LOG2": "31"}),
                        ("chunk 2^17", {"MSM_HIP_STREAM_MIN_LOG2": "18", "MSM_HIP_STREAM_CHUNK_LOG2": "17"}),
                        ("chunk 2^18", {"MSM_HIP_STREAM_MIN_LOG2": "18", "MSM_HIP_STREAM_CHUNK_LOG2": "18"})]
        if lg >= 20:
            configs += [("chunk 2^19", {"MSM_HIP_STREAM_MIN_LOG2": "18", "MSM_HIP_STREAM_CHUNK_LOG2": "19"})]
        if lg >= 22:
            configs += [("chunk 2^20", {"MSM_HIP_STREAM_MIN_LOG2": "18", "MSM_HIP_STREAM_CHUNK_LOG2": "20"})]
        for label, env in configs:
            for kind, b, s in (("pinned  ", hbp, hsp), ("pageable", hb, hs)):
                for stage in ("",):
                    e = dict(env)
                    os.environ.update(e)
                    try:
                        with mh.MsmContext() as c:
                            med, mn = timed(lambda: c.msm(b, s, mh.FORM_MONT))
                            tm = c.timings()
                    finally:
                        for k in e: os.environ.pop(k, None)
                    print(f"2^{lg} {kind} {label:12s} median {med:7.3f} min {mn:7.3f} ms  chunks {tm['stream_chunks']}", flush=True)
        with mh.MsmContext() as c:
            med, mn = timed(lambda: c.msm_arkworks(img, 72, 0, 32, 64, hs))
            print(f"2^{lg} arkworks pageable auto         median {med:7.3f} min {mn:7.3f} ms  chunks {c.timings()['stream_chunks']}", flush=True)
    gen.close()


if __name__ == "__main__":
    main()
