"""Device-resident MSM above 2^23 points: whole-instance sort vs point ranges of 2^22 / 2^21 into shared buckets (MSM_HIP_DEVICE_CHUNK_LOG2,
read once per process: one subprocess per setting).  usage: device_chunk_ab.py [log_n ...]"""
import os, subprocess, sys
os.environ.setdefault("MSM_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpu-acceleration_amd", "libmsm_hip_hooks.so"))  # the A/B knobs this script sets are read by the HOOKS build only (round 5)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path[:0] = [ROOT, os.path.join(ROOT, "gpu-acceleration_amd")]
    import time, torch
    import mopro_msm_hip as mh
    from mopro_msm_hip import testhooks as th
    lg = int(sys.argv[2]); n = 1 << lg
    hk = th.HooksContext()
    d_b = torch.empty(n * 16, dtype=torch.int32, device="cuda:0"); d_s = torch.empty(n * 8, dtype=torch.int32, device="cuda:0")
    hk.generate_device(1, 2, n, d_b.data_ptr(), d_s.data_ptr()); torch.cuda.synchronize()
    c = mh.MsmContext()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3: r = c.msm_device(d_b.data_ptr(), d_s.data_ptr(), n)
    ts = []
    for _ in range(9):
        t = time.perf_counter(); r = c.msm_device(d_b.data_ptr(), d_s.data_ptr(), n); ts.append((time.perf_counter() - t) * 1e3)
    ts.sort()
    print(f"median {ts[4]:.3f} min {ts[0]:.3f} ms  chunks {c.timings()['stream_chunks']}  x0 {int(r.affine_std[0]):08x}")
    sys.exit(0)
for lg in [int(a) for a in sys.argv[1:]] or [23, 24]:
    for knob in (os.environ.get("KNOBS") or "0,22,21").split(","):
        env = dict(os.environ, MSM_HIP_DEVICE_CHUNK_LOG2=knob)
        p = subprocess.run([sys.executable, __file__, "--child", str(lg)], env=env, capture_output=True, text=True)
        print(f"2^{lg} MSM_HIP_DEVICE_CHUNK_LOG2={knob:2s}: {p.stdout.strip() or p.stderr[-300:]}", flush=True)
