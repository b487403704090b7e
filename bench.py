#!/usr/bin/env python3
"""bench.py -- BN254 G1 MSM latency (ms) at N=2^20 on 1/2/4/8 MI355X (BASELINE.json metric).

A "step" is one complete MSM over the whole N=2^20 instance: each rank runs the HIP pipeline on its
contiguous point range (inputs already resident in its HBM), the per-rank partial group elements
(96 bytes each) are exchanged with one RCCL all-gather, and every rank folds them in rank order.
Total work is fixed as the GPU count grows => "scaling": "strong".  value = ms per step, max over ranks.

Launch: python bench.py [--gpus N --steps K --warmup W]
        python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...
Prints ONE JSON line on rank 0.  The oracle (oracle/) is used only for the cpu_baseline leg and as a
checker outside the timed region.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "gpu-acceleration_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

LOG_N = 20
BASE_SEED, SCALAR_SEED = 0xB2540001, 0xB2540002
STREAM_MUL = 0xD1342543DE82EF95  # element i of a stream = SplitMix64 seeded with seed + i*STREAM_MUL
HBM_PEAK_GBPS = 8000.0           # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
R_ORDER = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001


def words_to_ints(a):
    a = np.ascontiguousarray(a, dtype=np.uint32).reshape(-1, 8)
    out = [0] * a.shape[0]
    cols = [a[:, j].tolist() for j in range(8)]
    for i in range(a.shape[0]):
        v = 0
        for j in range(7, -1, -1):
            v = (v << 32) | cols[j][i]
        out[i] = v
    return out


# one mixed addition (ec_bn254.hpp xyzz_madd): 6 fp_mul + 2 fp_sqr + 1 fused fp_mul_add
MADS_PER_ADD = 6 * 162 + 2 * 126 + 243           # v_mad_u64_u32 instructions
FPMUL_EQ_PER_ADD = (6 * 171 + 2 * 135 + 252) / 171.0  # in units of one fp_mul (162 mads + 9 Montgomery-digit multiplications)


def valu_roofline(num_adds, acc_ms, mad_peak, fpmul_peak):
    if not (num_adds and acc_ms > 0 and mad_peak > 0 and fpmul_peak > 0):
        return None
    t = acc_ms * 1e-3
    mads = num_adds * MADS_PER_ADD / t
    fpm = num_adds * FPMUL_EQ_PER_ADD / t
    return {"bound": "valu", "kernel": "k_accumulate", "mixed_additions_per_launch": num_adds,
            "achieved": round(fpm / 1e9, 2), "peak": round(fpmul_peak / 1e9, 2), "unit": "G field-mul/s", "frac": round(fpm / fpmul_peak, 4),
            "mad_u64_achieved_G_per_s": round(mads / 1e9, 1), "mad_u64_peak_G_per_s": round(mad_peak / 1e9, 1),
            "mad_u64_frac": round(mads / mad_peak, 4),
            "note": "peaks measured live by msm_calibrate (csrc k_calibrate); field-mul = 9x29-bit Montgomery multiplication, "
                    "%.2f multiplication-equivalents per mixed addition" % FPMUL_EQ_PER_ADD}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log-n", type=int, default=LOG_N, help="total instance size (default 2^20, the BASELINE metric)")
    ap.add_argument("--window-bits", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-glv", action="store_true", help="A/B: run the unsplit pipeline (MSM_FLAG_NO_GLV)")
    ap.add_argument("--debug-same-device", action="store_true",
                    help="functional check of the N>1 path on a 1-GPU box: every rank uses cuda:0 and the exchange runs over gloo")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import mopro_msm_hip as mh

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with torch.distributed.run --nproc-per-node N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the MSM engine has no CPU fallback")
    if args.debug_same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    xdev = None if args.debug_same_device else dev  # device the 96-byte partials are exchanged on
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.debug_same_device:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)  # RCCL on ROCm

    n_total = 1 << args.log_n
    lo = rank * n_total // world
    hi = (rank + 1) * n_total // world
    n_local = hi - lo

    ctx_flags = mh.FLAG_NO_GLV if args.no_glv else 0
    ctx = mh.MsmContext(device=local_rank, window_bits=args.window_bits, flags=ctx_flags, max_points=n_local)
    d_bases = torch.empty(n_local * 16, dtype=torch.int32, device=dev)
    d_scalars = torch.empty(n_local * 8, dtype=torch.int32, device=dev)
    mask = (1 << 64) - 1
    ctx.generate_device((BASE_SEED + lo * STREAM_MUL) & mask, (SCALAR_SEED + lo * STREAM_MUL) & mask, n_local,
                        d_bases.data_ptr(), d_scalars.data_ptr())
    torch.cuda.synchronize()

    from mopro_msm_hip import distributed as md

    def step():
        # HIP pipeline on this rank's shard, then (N > 1) the exchange step: EC addition is not an RCCL
        # reduction op, so the "all-reduce" of partial group elements is an all-gather of 96 bytes per rank
        # over RCCL + a local fold in rank order (identical on all ranks)
        return md.distributed_msm_device(ctx, d_bases.data_ptr(), d_scalars.data_ptr(), n_local, device=xdev)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        res = step()
    ctx.reset_kernel_stats()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    fence()
    elapsed = time.perf_counter() - t0
    acc_avg_ms, acc_launches = ctx.accumulate_kernel_stats()
    mad_peak, fpmul_peak = ctx.calibrate() if rank == 0 else (0.0, 0.0)  # two ~1 ms micro-kernels, outside the timed region
    # per-stage hipEvents are off in the timed region (each record costs ~6 us of stream time): one extra, untimed
    # step with them on gives the stage breakdown
    ctx.set_stage_timing(True)
    step()
    tm = ctx.timings()
    ctx.set_stage_timing(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=xdev if xdev is not None else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed * 1e3 / args.steps

    # ---- correctness gate (outside the timed region): closed form (sum s_i k_i mod r) * G ----------
    k_loc = mh.generate_scalars_host((BASE_SEED + lo * STREAM_MUL) & mask, n_local, nonzero=True)
    s_loc = mh.generate_scalars_host((SCALAR_SEED + lo * STREAM_MUL) & mask, n_local)
    dot = sum(a * b for a, b in zip(words_to_ints(k_loc), words_to_ints(s_loc))) % R_ORDER
    if world > 1:
        dots = [None] * world
        dist.all_gather_object(dots, dot)
        dot = sum(dots) % R_ORDER

    if rank == 0:
        from oracle import bn254_oracle as orc  # checker + cpu_baseline leg only
        g = np.zeros(16, np.uint32)
        g[0], g[8] = 1, 2
        exp, exp_inf = orc.g1_to_affine_std(orc.g1_scalar_mul(g, orc.int_to_words(dot)))
        bit_exact = bool((res.affine_std == exp).all() and res.is_infinity == bool(exp_inf))

        pl = mh.plan(n_local, args.window_bits, ctx_flags)
        W, H = pl.num_windows, pl.num_buckets
        # ALGORITHMIC bytes of one accumulate launch: W*(N*(4 B index + 64 B affine point) + H*96 B)
        # (with the GLV split a window sorts and accumulates 2n virtual points in half as many windows: same point term)
        alg_bytes = W * (int(pl.virtual_points) * 68 + H * 96)
        achieved = alg_bytes / (acc_avg_ms * 1e-3) / 1e9 if acc_avg_ms > 0 else 0.0
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "accumulate_pmc.json")
        if os.path.exists(pmc):
            try:
                j = json.load(open(pmc))
                if j.get("n_local") == n_local and j.get("window_bits") == pl.window_bits:
                    traffic = j.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "BN254 G1 MSM latency (ms) at N=2^%d, bit-exact vs arkworks-equivalent oracle" % args.log_n,
            "value": round(ms_per_step, 4), "unit": "ms", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": False, "scaling": "strong", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": "BN254 G1 variable-base MSM, N=2^%d, dynamic window + signed-digit buckets "
                                   "(BASELINE.json configs[2]); bases k_i*G and scalars resident in HBM, "
                                   "point-range shards + all-gather of 96-byte partials" % args.log_n,
                       "n_total": n_total, "n_per_gpu": n_local, "window_bits": pl.window_bits, "num_windows": W,
                       "buckets_per_window": H, "glv_split": bool(pl.glv), "points_per_window": int(pl.virtual_points),
                       "parallelism": "point-range x%d" % world},
            "bit_exact": bit_exact,
            "roofline": {"bound": "hbm", "kernel": "k_accumulate", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic,
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_kernel_ms": round(acc_avg_ms, 4),
                         "launches_timed": int(acc_launches),
                         "note": "integer-multiply (VALU) bound kernel; HBM fraction reported because BASELINE.json asks for it; "
                                 "the multiplier roofline is in roofline_valu"},
            # the bound that actually holds: multiplier work of the launch against what two calibration
            # micro-kernels sustain on THIS device (dependent chains, 4 wavefronts per SIMD, like k_accumulate)
            "roofline_valu": valu_roofline(int(tm.get("num_adds", 0)), acc_avg_ms, mad_peak, fpmul_peak),
            "stage_ms_untimed_diagnostic_step": {k: round(v, 4) for k, v in tm.items() if k.endswith("_ms")},
        }
        if not args.no_cpu_baseline and world == 1:
            # CPU baseline: the arkworks-0.4-algorithm restatement (oracle_msm_pippenger) on this host's cores,
            # on the SAME bases/scalars (whole instance when it fits ~30 s of CPU work, else a prefix)
            threads = orc.threads_available()
            n_cpu = min(n_local, (1 << 20) if threads >= 8 else (1 << 18))
            hb = d_bases[: n_cpu * 16].cpu().numpy().view(np.uint32).reshape(n_cpu, 16)
            hs = d_scalars[: n_cpu * 8].cpu().numpy().view(np.uint32).reshape(n_cpu, 8)
            t0 = time.perf_counter()
            cpu_aff, cpu_inf, _ = orc.msm_pippenger(hb, hs, orc.FORM_MONT, None, threads)
            cpu_ms = (time.perf_counter() - t0) * 1e3
            if n_cpu == n_local:
                cpu_ok = bool((cpu_aff == res.affine_std).all())
            else:
                chk = ctx.msm_device(d_bases.data_ptr(), d_scalars.data_ptr(), n_cpu)
                cpu_ok = bool((cpu_aff == chk.affine_std).all())
            # arkworks parallelises over windows only (c = ln(n)*0.69 + 2 bits => 17 windows at 2^20): that many threads do work
            lg = int(np.log2(n_cpu))
            c_ark = 3 if n_cpu < 32 else (lg * 69) // 100 + 2
            busy = min(threads, -(-254 // c_ark))
            out["cpu_baseline"] = {"value": round(cpu_ms, 2), "unit": "ms", "cores": busy, "kind": "port",
                                   "sample": "first 2^%d points of the same instance, one MSM; arkworks-0.4 algorithm restated in C "
                                             "(not arkworks itself): one thread per window, %d windows of %d bits, %d host threads available"
                                             % (lg, -(-254 // c_ark), c_ark, threads),
                                   "agrees_with_gpu": cpu_ok}
        print(json.dumps(out))
        sys.stdout.flush()
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
