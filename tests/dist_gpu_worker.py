"""Worker of tests/test_gpu_3_configs.py::test_two_ranks_on_one_gpu (launched by torch.distributed.run, 2 processes, both on
cuda:0, exchange over gloo): rank r generates and runs its point range of one 2^18-point instance through
distributed_msm_device, then EVERY rank checks its own bits against the closed form and prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gpu-acceleration_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402


def main():
    import torch
    import torch.distributed as dist
    import mopro_msm_hip as mh
    from mopro_msm_hip import distributed as md
    from mopro_msm_hip import testhooks as th
    from oracle import bn254_oracle as orc

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    # MSM_TEST_BACKEND=nccl (tests/test_gpu_4_multi_device.py, boxes with >= 2 GPUs): one DEVICE per rank, exchange over RCCL/xGMI --
    # what bench.py --gpus N runs; default: every rank on cuda:0, exchange over gloo (1-GPU boxes)
    nccl = os.environ.get("MSM_TEST_BACKEND", "gloo") == "nccl"
    dev_index = int(os.environ.get("LOCAL_RANK", "0")) if nccl else 0
    torch.cuda.set_device(dev_index)
    if nccl:
        dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
    else:
        dist.init_process_group("gloo")
    xdev = torch.device("cuda", dev_index) if nccl else None  # device the 96-byte partials are exchanged on
    n_total = 1 << int(os.environ.get("MSM_TEST_LOG_N", "18"))
    lo, hi = md.shard_range(n_total, rank, world)
    n = hi - lo
    mul, mask = 0xD1342543DE82EF95, (1 << 64) - 1
    bs, ss = (0xB2540091 + lo * mul) & mask, (0xB2540092 + lo * mul) & mask
    d_b = torch.empty(n * 16, dtype=torch.int32, device=torch.device("cuda", dev_index))
    d_s = torch.empty(n * 8, dtype=torch.int32, device=torch.device("cuda", dev_index))
    with th.HooksContext(device=dev_index) as gen:
        gen.generate_device(bs, ss, n, d_b.data_ptr(), d_s.data_ptr())
    torch.cuda.synchronize()
    bad_rank = int(os.environ.get("MSM_TEST_BAD_SCALAR_RANK", "-1"))
    if bad_rank >= 0:
        # failure propagation: ONE rank's shard holds a scalar >= 2^254; every rank must raise MsmError(ERR_BAD_ARG), nobody hangs
        if rank == bad_rank:
            d_s[8 * (n // 3) + 7] = 0x40000000
            torch.cuda.synchronize()
        code, good, msg = None, False, ""
        with mh.MsmContext(device=dev_index) as ctx:
            for _ in range(2):
                try:
                    md.distributed_msm_device(ctx, d_b.data_ptr(), d_s.data_ptr(), n, device=xdev)
                except mh.MsmError as e:
                    code = e.code
                    msg = str(e)
            good = code == mh.ERR_BAD_ARG and ("rank %d" % bad_rank) in msg
        sys.stdout.write(json.dumps({"rank": rank, "ok": good, "error_code": code, "message": msg}) + "\n")
        sys.stdout.flush()
        dist.barrier()
        dist.destroy_process_group()
        sys.exit(0 if good else 1)
    # MSM_TEST_DETERMINISTIC=1: the contexts carry MSM_FLAG_DETERMINISTIC, so the fold of the ranks' partials must hand out the
    # canonical Z = 1 Jacobian words -- identical on every rank and on every repetition, Z == R mod p (Montgomery one)
    det = os.environ.get("MSM_TEST_DETERMINISTIC", "0") == "1"
    jac_seen = set()
    with mh.MsmContext(device=dev_index, flags=mh.FLAG_DETERMINISTIC if det else 0) as ctx:
        res = None
        for _ in range(3):
            res = md.distributed_msm_device(ctx, d_b.data_ptr(), d_s.data_ptr(), n, device=xdev)
            jac_seen.add(bytes(res.jacobian_mont.tobytes()))
    rccl_single = None
    if nccl and world == 1:
        # ONE rank over the real RCCL backend (round 5; 1-GPU boxes): all_reduce_msm returns early for world 1, so the exchange itself --
        # pinned staging, 100 B up, all_gather_into_tensor on the rank's device, 100 B down, stream synchronisation, fold -- is run directly
        parts, status = md._exchange(xdev, None).run(res.jacobian_mont, mh.OK, None)
        folded = mh.combine_partials(parts)
        rccl_single = bool(parts.shape == (1, 24) and int(status[0]) == mh.OK and (parts[0] == res.jacobian_mont).all()
                           and (folded.affine_std == res.affine_std).all() and md._exchange(xdev, None).last_ms > 0)
    k = th.generate_scalars_host(bs, n, nonzero=True)
    s = th.generate_scalars_host(ss, n)
    to_int = lambda a: [sum(int(w) << (32 * j) for j, w in enumerate(row)) for row in a.tolist()]
    dot = sum(a * b for a, b in zip(to_int(k), to_int(s)))
    dots = [None] * world
    dist.all_gather_object(dots, dot)
    g = np.zeros(16, np.uint32)
    g[0], g[8] = 1, 2
    exp, einf = orc.g1_to_affine_std(orc.g1_scalar_mul(g, orc.int_to_words(sum(dots) % orc.R_ORDER)))
    ok = bool((res.affine_std == exp).all()) and not res.is_infinity and rccl_single is not False
    if det:
        one = orc.fq_to_mont(orc.int_to_words(1))
        ok = ok and len(jac_seen) == 1 and bool((res.jacobian_mont[16:24] == one).all())
    sys.stdout.write(json.dumps({"rank": rank, "ok": ok, "affine": res.affine_std.tolist(), "device": dev_index,
                                 "jacobian": res.jacobian_mont.tolist(), "jacobian_representations": len(jac_seen),
                                 "backend": "nccl" if nccl else "gloo", "rccl_exchange_with_one_rank": rccl_single}) + "\n")  # ONE write: the ranks share a pipe
    sys.stdout.flush()
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
