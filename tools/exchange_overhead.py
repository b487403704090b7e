#!/usr/bin/env python3
"""Per-step cost of the multi-GPU exchange (all-gather of 96-byte partials over RCCL + fold), measured with a
single-rank "nccl" group on one GPU: everything but the inter-GPU hop itself (launch, staging copies, sync, host fold)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gpu-acceleration_amd")]
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import numpy as np, torch, torch.distributed as dist
import mopro_msm_hip as mh
from mopro_msm_hip import testhooks as th
GEN = th.HooksContext()  # the synthetic-instance generator lives in the hooks build
from mopro_msm_hip import distributed as md
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
n = 1 << 14
ctx = mh.MsmContext(max_points=n)
d_b = torch.empty(n * 16, dtype=torch.int32, device=dev); d_s = torch.empty(n * 8, dtype=torch.int32, device=dev)
GEN.generate_device(1, 2, n, d_b.data_ptr(), d_s.data_ptr()); torch.cuda.synchronize()
r = ctx.msm_device(d_b.data_ptr(), d_s.data_ptr(), n)

def old(partial):
    mine = torch.from_numpy(np.ascontiguousarray(partial, dtype=np.uint32).view(np.int32).copy()).to(dev)
    out = torch.empty(24, dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(out, mine)
    return out.cpu().numpy().view(np.uint32).reshape(1, 24)

for name, fn in (("old (allocating, pageable)", old), ("new (reused pinned buffers)", lambda p: md.all_gather_partials(p, dev))):
    for _ in range(20): fn(r.jacobian_mont)
    ts = []
    for _ in range(300):
        t0 = time.perf_counter(); g = fn(r.jacobian_mont); ts.append((time.perf_counter() - t0) * 1e6)
    ts.sort()
    t0 = time.perf_counter()
    for _ in range(300): mh.combine_partials(np.repeat(g, 8, axis=0))
    tc = (time.perf_counter() - t0) / 300 * 1e6
    print(f"{name}: exchange median {ts[150]:.1f} us, p90 {ts[270]:.1f} us; fold of 8 partials {tc:.1f} us")
dist.destroy_process_group()
