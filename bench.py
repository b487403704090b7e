#!/usr/bin/env python3
"""bench.py -- BN254 G1 MSM latency (ms) at N=2^20 on 1/2/4/8 MI355X (BASELINE.json metric).

A "step" is one complete MSM over the whole N=2^20 instance: each rank runs the HIP pipeline on its
contiguous point range (inputs already resident in its HBM), the per-rank partial group elements
(96 bytes each) are exchanged with one RCCL all-gather, and every rank folds them in rank order.
Total work is fixed as the GPU count grows => "scaling": "strong".  value = ms per step, max over ranks.

Launch: python bench.py [--gpus N --steps K --warmup W]
        python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...
        python bench.py --gpus N --in-process      (one process, msm_multi: a context + host thread per device, in-library RCCL)
        ... --log-n 24            BASELINE config 4 (2^21 points per GPU on 8 GPUs)
        ... --log-n 26 --streamed BASELINE config 5 (host->HBM chunks overlapped with the accumulation; 2^23 points per GPU on 8)
Prints ONE JSON line on rank 0 and exits non-zero if the result of the last timed step is not bit-exact.  The oracle
(oracle/) is used only for the cpu_baseline leg and as a checker outside the timed region; inputs come from the hooks build
(libmsm_hip_hooks.so: generator, calibration), the timed path is the product library (libmsm_hip.so).
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
MASK64 = (1 << 64) - 1
LAYOUT_NAMES = {1: "one stream", 2: "one stream + reduce stream", 3: "two streams"}  # MSM_BATCH_LAYOUT_*


def dot_words(k_words, s_words):
    """sum_i k_i * s_i as a Python integer for two (n, 8) arrays of little-endian u32 words: 16-bit halves, float64 BLAS in slices of 2^20
    (every partial sum < 2^32 * 2^20 = 2^52: exact) -- the closed-form check of a 2^26-point instance takes seconds, not minutes"""
    k = np.ascontiguousarray(k_words, dtype=np.uint32).reshape(-1, 8)
    s = np.ascontiguousarray(s_words, dtype=np.uint32).reshape(-1, 8)
    tot = 0
    for lo in range(0, min(len(k), len(s)), 1 << 20):
        k16 = k[lo:lo + (1 << 20)].view(np.uint16).reshape(-1, 16).astype(np.float64)
        s16 = s[lo:lo + (1 << 20)].view(np.uint16).reshape(-1, 16).astype(np.float64)
        m = k16.T @ s16
        for a in range(16):
            for b in range(16):
                tot += int(m[a, b]) << (16 * (a + b))
    return tot


def kernel_source_hash():
    """sha256 (16 hex digits) over the kernel / host-runtime sources the product library is built from: the committed counter passes
    (profiles/*_pmc_*.json) carry the hash of the build they measured, and a line that quotes them for a different build says so"""
    import glob
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "gpu-acceleration_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.hpp")) + glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.inc"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


EXTRAS_DEADLINE_S = int(os.environ.get("MSM_BENCH_EXTRAS_DEADLINE_S", "420"))  # the untimed legs of a default run take ~6 s

# one mixed addition (ec_bn254.hpp xyzz_madd, the loop body of k_accumulate_pieces): 6 fp_mul + 2 fp_sqr + 1 fused fp_mul_add
MADS_PER_ADD = 6 * 162 + 2 * 126 + 243           # v_mad_u64_u32 instructions
FPMUL_EQ_PER_ADD = (6 * 171 + 2 * 135 + 252) / 171.0  # in units of one fp_mul (162 mads + 9 Montgomery-digit multiplications)


# vector instructions of one mixed addition in k_accumulate_pieces<.., M256>'s loop (disassembly of the round-5 build: 1468 v_mad_u64_u32, 81
# v_mul_lo_u32, 144 v_lshrrev_b64, 194 v_and_b32, the rest adds / subs / unpack / the digit's sign: 2117 in the loop body + 9 under the sign's
# lane mask; rocprofv3 --pmc counted 2101 per addition and lane for the round-4 loop of 2126, profiles/accumulate_valu_pmc.json)
VALU_INSTS_PER_ADD = 2126
# the same stream priced class by class at the stand-alone rates of profiles/microbench_r1_instruction_rates.txt (DESIGN.md section 4.1,
# yardstick i): multiplier instructions 4.8 cycles, 64-bit shifts / alignbit 4.5, VOP2 masks / adds / shifts 2.4
CLASS_PRICED_CYCLES_PER_ADD = (1468 + 81) * 4.8 + (144 + 23 + 2 + 1) * 4.5 + (VALU_INSTS_PER_ADD - 1468 - 81 - 170) * 2.4


def issue_floor(num_adds, mcycles, simds):
    """two YARDSTICKS for the kernel's instruction stream (neither is a hard bound: VOP2 instructions issue faster than 4 cycles when
    several wavefronts share a SIMD, multiplier instructions slower)"""
    if not (num_adds and mcycles and simds):
        return None
    wave_adds_per_simd = num_adds / 64.0 / simds
    cpi = mcycles * 1e6 / (wave_adds_per_simd * VALU_INSTS_PER_ADD)
    priced = wave_adds_per_simd * CLASS_PRICED_CYCLES_PER_ADD / 1e6
    return {"vector_instructions_per_addition": VALU_INSTS_PER_ADD, "simds": simds, "wavefront_additions_per_simd": round(wave_adds_per_simd, 1),
            "floor_mcycles": round(wave_adds_per_simd * VALU_INSTS_PER_ADD * 4 / 1e6, 3), "measured_mcycles": mcycles,
            "cycles_per_instruction": round(cpi, 3), "frac_of_issue_floor": round(4.0 / cpi, 4),
            "class_priced_mcycles": round(priced, 3), "measured_over_class_priced": round(mcycles / priced, 4),
            "note": "YARDSTICKS, not bounds.  floor_mcycles = 4 cycles per vector instruction and wavefront on a 16-lane SIMD (VOP2 adds / masks issue "
                    "faster with several wavefronts per SIMD: 2.2-2.6 cycles measured; the guide lists v_fma_f32 wave64 at 2); class_priced_mcycles = every "
                    "instruction class at its measured stand-alone rate (multiplier 4.8, 64-bit shifts 4.5, VOP2 2.4 cycles; profiles/"
                    "microbench_r1_instruction_rates.txt).  measured = live kernel ms x the shader clock the kernel measured for itself (clock.k_accumulate_mcycles)"}


def valu_roofline(num_adds, acc_ms, mad_peak, fpmul_peak):
    if not (num_adds and acc_ms > 0 and mad_peak > 0 and fpmul_peak > 0):
        return None
    t = acc_ms * 1e-3
    mads = num_adds * MADS_PER_ADD / t
    fpm = num_adds * FPMUL_EQ_PER_ADD / t
    out = {"bound": "valu", "kernel": "k_accumulate_pieces", "mixed_additions_per_launch": num_adds,
           "achieved": round(fpm / 1e9, 2), "peak": round(fpmul_peak / 1e9, 2), "unit": "G field-mul/s", "frac": round(fpm / fpmul_peak, 4),
           "mad_u64_achieved_G_per_s": round(mads / 1e9, 1), "mad_u64_peak_G_per_s": round(mad_peak / 1e9, 1),
           "mad_u64_frac": round(mads / mad_peak, 4),
           "note": "peaks measured live by msm_calibrate (hooks build, k_calibrate); field-mul = 9x29-bit Montgomery multiplication, "
                   "%.2f multiplication-equivalents per mixed addition" % FPMUL_EQ_PER_ADD}
    vb = os.path.join(ROOT, "profiles", "accumulate_valu_pmc.json")
    if os.path.exists(vb):  # hardware VALU-busy counters of the same kernel, collected offline (tools/pmc_valu.sh)
        try:
            j = json.load(open(vb))
            out["valu_busy_counters"] = {k: j[k] for k in ("valu_busy_frac", "valu_util_frac", "valu_insts_per_mixed_addition", "build", "source") if k in j}
        except Exception:
            pass
    return out


def timed_calls(fn, reps, warm=1, warm_s=0.1):
    """(result, median ms, min ms) of `reps` calls after untimed ones (at least `warm`, and for at least `warm_s` seconds: the GPU
    clock drops after ~50 ms without work and needs ~35 ms of work to come back, tools/clock_ramp.py)"""
    t_w = time.perf_counter()
    k = 0
    while k < warm or time.perf_counter() - t_w < warm_s:
        r = fn()
        k += 1
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        r = fn()
        ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    return r, ts[len(ts) // 2], ts[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--pre-warm-ms", type=float, default=150.0,
                    help="untimed steps run for this long BEFORE the warm-up steps: after ~50 ms without work the GPU clock drops and takes "
                         "~20 MSMs (35 ms) to come back (tools/clock_ramp.py: steps 1-5 after an idle gap 2.0-2.4 ms, steady state 1.67)")
    ap.add_argument("--kernel-timing-every", type=int, default=1,
                    help="every n-th launch of k_accumulate_pieces in the timed loop carries its pair of hipEvents (the roofline's live kernel time); a "
                         "timed dispatch does not overlap its neighbours' launch latency (~11 us per MSM), the library's default is none (msm_set_kernel_timing)")
    ap.add_argument("--log-n", type=int, default=LOG_N, help="total instance size (default 2^20, the BASELINE metric; 24 = config 4, 26 --streamed = config 5)")
    ap.add_argument("--window-bits", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-legs", action="store_true", help="skip the untimed host-pointer legs (pinned / pageable / arkworks zero-copy)")
    ap.add_argument("--no-glv", action="store_true", help="A/B: run the unsplit pipeline (MSM_FLAG_NO_GLV)")
    ap.add_argument("--streamed", action="store_true",
                    help="BASELINE config 5: the timed step is the HOST-pointer call on pinned caller memory (host->HBM chunks overlapped "
                         "with the accumulation) instead of the resident call; value then includes PCIe")
    ap.add_argument("--in-process", action="store_true",
                    help="N GPUs driven by ONE process through msm_multi (context + host thread per device, in-library RCCL exchange)")
    ap.add_argument("--exchange", choices=["auto", "rccl", "host"], default="auto",
                    help="--in-process only: the exchange of msm_multi (MSM_MULTI_EXCHANGE_*): auto = measured when the handle is created, "
                         "rccl = in-library ncclAllGather, host = the calling thread folds the partials")
    ap.add_argument("--debug-same-device", action="store_true",
                    help="functional check of the N>1 path on a 1-GPU box: every rank uses cuda:0 and the exchange runs over gloo / the host fold")
    ap.add_argument("--master-port", type=int, default=0, help="rendezvous port of the self-launched torchrun (0 = a free one)")
    args = ap.parse_args()

    # `python bench.py --gpus N` started plainly (no torchrun around it): start the N ranks ourselves -- as a CHILD process, before
    # anything here has touched the GPU (never a re-exec) -- relay rank 0's JSON line and exit with the child's code.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.in_process:
        import socket
        import subprocess
        port = args.master_port
        if not port:
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        sys.exit(subprocess.call(cmd, env=env))  # (no watchdog in this parent: exiting it would orphan the ranks)

    import faulthandler
    faulthandler.dump_traceback_later(1500, exit=True)  # a stalled run leaves the stacks of all threads on stderr instead of nothing
    import torch
    import torch.distributed as dist
    import mopro_msm_hip as mh
    from mopro_msm_hip import testhooks as th

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and not (world == 1 and args.in_process):
        raise SystemExit("WORLD_SIZE %d does not match --gpus %d" % (world, args.gpus))
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
    ctx_flags = mh.FLAG_NO_GLV if args.no_glv else 0
    in_proc = args.in_process and world == 1 and args.gpus >= 1
    nshards = args.gpus if in_proc else world

    # ---- inputs: shard g = points [g*N/G, (g+1)*N/G) generated on the device that owns it -------------------------------
    def shard(g):
        return g * n_total // nshards, (g + 1) * n_total // nshards

    my_shards = list(range(nshards)) if in_proc else [rank]
    devs = {g: (torch.device("cuda", 0 if args.debug_same_device else g) if in_proc else dev) for g in my_shards}
    d_bases, d_scalars = {}, {}
    for g in my_shards:
        lo, hi = shard(g)
        with th.HooksContext(device=devs[g].index) as gen:
            d_bases[g] = torch.empty((hi - lo) * 16, dtype=torch.int32, device=devs[g])
            d_scalars[g] = torch.empty((hi - lo) * 8, dtype=torch.int32, device=devs[g])
            gen.generate_device((BASE_SEED + lo * STREAM_MUL) & MASK64, (SCALAR_SEED + lo * STREAM_MUL) & MASK64, hi - lo,
                                d_bases[g].data_ptr(), d_scalars[g].data_ptr())
    torch.cuda.synchronize()
    lo, hi = shard(my_shards[0])
    n_local = hi - lo

    from mopro_msm_hip import distributed as md

    multi = None
    if in_proc:
        multi = mh.MsmMulti(devices=[devs[g].index for g in my_shards], window_bits=args.window_bits, flags=ctx_flags,
                            exchange={"auto": mh.EXCHANGE_AUTO, "rccl": mh.EXCHANGE_RCCL, "host": mh.EXCHANGE_HOST}[args.exchange])
        ctx = None
    else:
        ctx = mh.MsmContext(device=local_rank, window_bits=args.window_bits, flags=ctx_flags, max_points=n_local)

    hb_pin = hs_pin = None
    if args.streamed:  # config 5: the instance lives in (pinned) host memory; the device copies are dropped (2^26 points: 6.4 GB)
        g0 = rank if not in_proc else 0
        hb_pin = torch.empty(d_bases[g0].shape, dtype=torch.int32, pin_memory=True)
        hs_pin = torch.empty(d_scalars[g0].shape, dtype=torch.int32, pin_memory=True)
        hb_pin.copy_(d_bases[g0])
        hs_pin.copy_(d_scalars[g0])
        torch.cuda.synchronize()
        d_bases[g0] = d_scalars[g0] = None
        torch.cuda.empty_cache()
        hbn = hb_pin.numpy().view(np.uint32).reshape(-1, 16)
        hsn = hs_pin.numpy().view(np.uint32).reshape(-1, 8)

    shard_ms_acc = [0.0, 0]  # this rank's local MSM: summed wall clock, calls (reset before the timed loop)

    def local_msm():
        t_l = time.perf_counter()
        r = ctx.msm(hbn, hsn, mh.FORM_MONT) if args.streamed else ctx.msm_device(d_bases[rank].data_ptr(), d_scalars[rank].data_ptr(), n_local)
        shard_ms_acc[0] += (time.perf_counter() - t_l) * 1e3
        shard_ms_acc[1] += 1
        return r

    def step():
        # HIP pipeline on this rank's shard, then (N > 1) the exchange step: EC addition is not an RCCL
        # reduction op, so the "all-reduce" of partial group elements is an all-gather of 96 bytes (+ a status word) per rank
        # over RCCL + a local fold in rank order (identical on all ranks).  A rank whose local MSM fails still joins the
        # all-gather (md.guarded): every rank raises the first failing rank's code, nobody hangs.
        if in_proc:
            return multi.msm_device([d_bases[g].data_ptr() for g in my_shards], [d_scalars[g].data_ptr() for g in my_shards],
                                    [shard(g)[1] - shard(g)[0] for g in my_shards])
        return md.guarded(local_msm, xdev)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # clock ramp (untimed, before the warm-up steps): the instance generation, context creation and imports above leave the GPU
    # idle for tens of ms; the first ~20 MSMs after that run at a lower clock
    pre_warm_steps = 0
    if world > 1:  # step() holds a collective: the same count on every rank
        for _ in range(int(args.pre_warm_ms)):
            step()
            pre_warm_steps += 1
    else:
        t_pw = time.perf_counter()
        while (time.perf_counter() - t_pw) * 1e3 < args.pre_warm_ms:
            step()
            pre_warm_steps += 1
    for _ in range(args.warmup):
        res = step()
    if ctx is not None:
        ctx.set_kernel_timing(max(1, args.kernel_timing_every))
        ctx.reset_kernel_stats()
    else:
        multi.set_kernel_timing(1)  # (per-rank kernel times of the last step are read from msm_multi_get_timings)
    shard_ms_acc[0], shard_ms_acc[1] = 0.0, 0
    exch_ms_sum = 0.0
    step_jac = []  # the Jacobian words every timed step returned (96 bytes each): all checked after the loop
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
        step_jac.append(res.jacobian_mont)
        if world > 1:
            exch_ms_sum += md.last_exchange_ms(xdev)
    fence()
    elapsed = time.perf_counter() - t0
    clk_loop = clk_single = None
    if ctx is not None:
        clk_loop = ctx.clock_stats()  # k_accumulate_pieces' own cycle / constant-rate counters over the timed launches
        acc_avg_ms, acc_launches = ctx.accumulate_kernel_stats()
        # instances cut into point ranges (device inputs from 2^23 points, streamed host inputs) launch k_accumulate_pieces once per range:
        # the roofline prices the MSM's accumulation = all of a step's launches together, against the whole instance's bytes
        # (the context's statistic holds the LAST range's launch of every call; the ranges are equal: 2^22 points each)
        launches_per_step = 1 if args.streamed else max(1, int(ctx.timings().get("stream_chunks", 0)))
        acc_avg_ms *= launches_per_step
    else:
        acc_avg_ms, acc_launches, launches_per_step = multi.timings(0)["accumulate_ms"], 1, 1
    # what a caller sees who does NOT keep the GPU busy: one step after 0.2 s without work (the clock has dropped)
    after_idle_ms = None
    if world == 1 and not in_proc:
        time.sleep(0.2)
        t_i = time.perf_counter()
        step()
        after_idle_ms = (time.perf_counter() - t_i) * 1e3
    mad_peak, fpmul_peak = (0.0, 0.0)
    if rank == 0:  # two ~1 ms micro-kernels, outside the timed region
        with th.HooksContext(device=devs[my_shards[0]].index) as cal:
            # at the SAME clock state as the timed steps: the calibration kernels run back to back for ~100 ms and the best pair
            # counts (a single cold pair right after creating the context read 141 G field-mul/s where the ramped-up device does 162)
            t_c = time.perf_counter()
            while (time.perf_counter() - t_c) < 0.1:
                m_, f_ = cal.calibrate()
                mad_peak, fpmul_peak = max(mad_peak, m_), max(fpmul_peak, f_)
    # per-stage hipEvents are off in the timed region (each record costs ~6 us of stream time): one extra, untimed
    # step with them on gives the stage breakdown
    tm = {}
    if ctx is not None and not args.streamed:
        ctx.set_stage_timing(True)
        ctx.reset_kernel_stats()
        step()
        tm = ctx.timings()
        clk_single = ctx.clock_stats()
        ctx.set_stage_timing(False)
    elif ctx is not None:
        tm = ctx.timings()
    else:
        tm = multi.timings(0)
    # what makes the first multi-GPU run self-explaining (no 8-GPU node was available to the builder): the world size the exchange
    # really saw, what the exchange cost, and every rank's own shard time
    exchange = None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=xdev if xdev is not None else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        per_rank = [None] * world
        dist.all_gather_object(per_rank, {"rank": rank, "device": torch.cuda.current_device(), "shard_points": n_local,
                                          "shard_ms": shard_ms_acc[0] / max(1, shard_ms_acc[1]), "exchange_ms": exch_ms_sum / max(1, args.steps),
                                          "sclk_ghz": round(clk_loop["sclk_ghz"], 4) if clk_loop and clk_loop["samples"] else None,
                                          "k_accumulate_ms": round(acc_avg_ms, 4)})
        sh = [p["shard_ms"] for p in per_rank]
        exchange = {"backend": "gloo (debug: every rank on cuda:0)" if args.debug_same_device else "rccl (torch.distributed nccl backend)",
                    "world_seen": dist.get_world_size(), "devices_seen": sorted({p["device"] for p in per_rank}),
                    "payload_bytes_per_rank": 4 * md.WORDS, "ms_per_step": round(max(p["exchange_ms"] for p in per_rank), 4),
                    "ms_per_step_min_rank": round(min(p["exchange_ms"] for p in per_rank), 4),
                    "shard_ms_max": round(max(sh), 4), "shard_ms_min": round(min(sh), 4),
                    # a slow rank is a slow KERNEL or a slow CLOCK: every rank's own shader clock over its timed launches beside its shard time
                    "per_rank": [{"rank": p["rank"], "device": p["device"], "shard_ms": round(p["shard_ms"], 4), "exchange_ms": round(p["exchange_ms"], 4),
                                  "sclk_ghz": p["sclk_ghz"], "k_accumulate_ms": p["k_accumulate_ms"]} for p in per_rank],
                    "note": "exchange ms = host wall clock of the 100-byte all-gather incl. its two copies and the wait for the slowest rank"}
    elif in_proc:
        ex_ms, sh = multi.exchange_stats()
        p_rccl, p_host = multi.exchange_probe()
        exchange = {"backend": {1: "rccl (in-library, ncclCommInitAll)", 2: "host fold"}.get(multi.exchange, "?"),
                    "requested": args.exchange,
                    "auto_probe_ms": {"rccl": round(p_rccl, 4), "host": round(p_host, 4)} if (p_rccl or p_host) else None,
                    "world_seen": multi.num_devices, "devices_seen": sorted({devs[g].index for g in my_shards}),
                    "payload_bytes_per_rank": 96, "ms_per_step": round(ex_ms, 4), "shard_ms_max": round(max(sh), 4),
                    "shard_ms_min": round(min(sh), 4),
                    "per_rank": [{"rank": g, "device": devs[g].index, "shard_ms": round(sh[g], 4),
                                  "sclk_ghz": round(multi.clock_stats(g)["sclk_ghz"], 4), "k_accumulate_ms": round(multi.timings(g)["accumulate_ms"], 4)}
                                 for g in range(multi.num_devices)],
                    "note": "last step; shard ms = wall clock of each rank's local MSM on its host thread; sclk_ghz = the shader clock that rank's "
                            "k_accumulate_pieces measured for itself since the handle was created"}
    ms_per_step = elapsed * 1e3 / args.steps

    # ---- correctness gate (outside the timed region): closed form (sum s_i k_i mod r) * G ----------
    dot = 0
    for g in my_shards:
        glo, ghi = shard(g)
        for c0 in range(glo, ghi, 1 << 22):  # in slices: a 2^26-point shard never holds more than 2^22 x 64 bytes of logs
            cnt = min(1 << 22, ghi - c0)
            k_loc = th.generate_scalars_host((BASE_SEED + c0 * STREAM_MUL) & MASK64, cnt, nonzero=True)
            s_loc = th.generate_scalars_host((SCALAR_SEED + c0 * STREAM_MUL) & MASK64, cnt)
            dot += dot_words(k_loc, s_loc)
    dot %= R_ORDER
    dot_local = dot
    if world > 1:
        dots = [None] * world
        dist.all_gather_object(dots, dot)
        dot = sum(dots) % R_ORDER

    bit_exact = True
    if rank == 0:
        # the CPU leg's OpenMP threads stay on their cores (libgomp reads these when the oracle library is loaded)
        os.environ.setdefault("OMP_PROC_BIND", "close")
        os.environ.setdefault("OMP_PLACES", "cores")
        from oracle import bn254_oracle as orc  # checker + cpu_baseline leg only
        gpt = np.zeros(16, np.uint32)
        gpt[0], gpt[8] = 1, 2

        def expect(d):
            return orc.g1_to_affine_std(orc.g1_scalar_mul(gpt, orc.int_to_words(d % R_ORDER)))

        exp, exp_inf = expect(dot)
        bit_exact = bool((res.affine_std == exp).all() and res.is_infinity == bool(exp_inf))
        # EVERY timed step is checked: its Jacobian words are normalised (the one field inversion, outside the timed region) and the
        # affine words compared with the closed form.  The words themselves may differ from step to step -- the sort places the entries of a
        # bucket with LDS atomics, so the order of a bucket's additions, and with it the projective representation, is not fixed -- the
        # group element is.
        steps_ok = 0
        for jw in step_jac:
            a = mh.combine_partials(jw.reshape(1, 24))
            steps_ok += int(bool((a.affine_std == exp).all()) and a.is_infinity == bool(exp_inf))
        bit_exact = bit_exact and steps_ok == len(step_jac)

        pl = mh.plan(n_local, args.window_bits, ctx_flags)
        W, H = pl.num_windows, pl.num_buckets
        # ALGORITHMIC bytes of one accumulate launch: W*(N*(4 B index + 64 B affine point) + H*96 B)
        # (with the GLV split a window sorts and accumulates 2n virtual points in half as many windows: same point term)
        alg_bytes = W * (int(pl.virtual_points) * 68 + H * 96)
        achieved = alg_bytes / (acc_avg_ms * 1e-3) / 1e9 if acc_avg_ms > 0 and not args.streamed else 0.0
        # HBM traffic of one k_accumulate launch: PMC counters cannot be read from inside this process, so the figure comes from the
        # committed offline passes (tools/pmc_accumulate.sh: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs, corrected as the
        # microarchitecture guide prescribes) of the SAME shape -- profiles/accumulate_pmc*.json, one file per shard size -- and says so
        src_hash = kernel_source_hash()
        traffic, traffic_src = None, {"source": "none", "detail": "no committed PMC pass for n_local %d, c %d" % (n_local, pl.window_bits)}
        import glob
        for pmc in sorted(glob.glob(os.path.join(ROOT, "profiles", "accumulate_pmc*.json"))):
            try:
                j = json.load(open(pmc))
            except Exception:
                continue
            if j.get("n_local") == n_local and j.get("window_bits") == pl.window_bits and j.get("num_windows", W) == W:
                traffic = j.get("hbm_bytes_per_launch")
                traffic_src = {"source": "file", "file": os.path.relpath(pmc, ROOT), "build": j.get("build", "round 2"),
                               "source_hash_of_measured_build": j.get("source_hash"), "traffic_stale": j.get("source_hash") != src_hash,
                               "detail": "offline rocprofv3 --pmc passes on the same box class (tools/pmc_accumulate.sh); not measured in this run; "
                                         "traffic_stale = the kernel sources have changed since the counters were collected"}
                break
        sort_ms = float(tm.get("sort_ms", 0.0) or 0.0)
        sort_bytes = 8 * int(pl.virtual_points) * W  # per window N*(2 read + 2 read + 4 write)
        sort_traffic, sort_traffic_src = None, None  # HBM bytes the six sort launches really move (same offline PMC passes, tools/pmc_sort_summarize.py)
        for pmc in sorted(glob.glob(os.path.join(ROOT, "profiles", "sort_pmc*.json"))):
            try:
                j = json.load(open(pmc))
            except Exception:
                continue
            if j.get("n_local") == n_local and j.get("window_bits") == pl.window_bits and bool(j.get("glv_split")) == bool(pl.glv):
                sort_traffic = j.get("sort_hbm_bytes")
                sort_traffic_src = {"source": "file", "file": os.path.relpath(pmc, ROOT), "build": j.get("build"),
                                    "source_hash_of_measured_build": j.get("source_hash"), "traffic_stale": j.get("source_hash") != src_hash,
                                    "detail": "offline rocprofv3 --pmc passes (tools/pmc_accumulate.sh + tools/pmc_sort_summarize.py); not measured in this run"}
                break
        out = {
            "metric": "BN254 G1 MSM latency (ms) at N=2^%d, bit-exact vs arkworks-equivalent oracle" % args.log_n,
            "value": round(ms_per_step, 4), "unit": "ms", "n_gpus": nshards, "steps": args.steps, "warmup": args.warmup, "pre_warm_steps": pre_warm_steps,
            "step_after_0.2s_idle_ms": round(after_idle_ms, 4) if after_idle_ms is not None else None,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": False, "scaling": "strong", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic", "source_hash": src_hash,
            "config": {"workload": "BN254 G1 variable-base MSM, N=2^%d, dynamic window + signed-digit buckets "
                                   "(BASELINE.json configs[%d]); bases k_i*G and scalars %s, "
                                   "point-range shards + all-gather of 96-byte partials"
                                   % (args.log_n, 4 if args.streamed else 3 if args.log_n >= 24 else 2,
                                      "in PINNED HOST memory, streamed host->HBM in chunks overlapped with the accumulation" if args.streamed
                                      else "resident in HBM"),
                       "n_total": n_total, "n_per_gpu": n_local, "window_bits": pl.window_bits, "num_windows": W,
                       "buckets_per_window": H, "glv_split": bool(pl.glv), "points_per_window": int(pl.virtual_points),
                       "parallelism": "point-range x%d%s" % (nshards, " (one process, msm_multi, exchange=%s)" %
                                                             {1: "rccl", 2: "host-fold"}.get(multi.exchange, "?") if in_proc else ""),
                       "timed_call": "msm_multi_device" if in_proc else "msm_bn254_g1 (host pointers, pinned)" if args.streamed else "msm_bn254_g1_device",
                       "timed_call_inputs": "pinned HOST memory: the timed call includes the PCIe transfer" if args.streamed else
                                            "HBM-resident: bases and scalars are in device memory when the timed call starts (no PCIe transfer inside `value`); "
                                            "the reference's own call shape -- host slices in, benches/e2e.rs:46-60 -- is value_host_pinned_ms / "
                                            "value_host_pageable_ms / value_host_arkworks_ms beside `value`",
                       "timed_call_returns": "Jacobian Montgomery words (the reference's own result type, metal_msm.rs:228-241); the one field "
                                             "inversion that gives the compared affine words (~10 us on the host) is outside the timed region"},
            "bit_exact": bit_exact,
            "bit_exact_steps": {"checked": len(step_jac), "equal_to_closed_form": steps_ok,
                                "distinct_jacobian_representations": len({bytes(jw.tobytes()) for jw in step_jac})},
            # Is a slower line a slower kernel or a slower box?  k_accumulate_pieces' first workgroup reads the shader-cycle counter and the
            # constant-rate counter around its chunk in EVERY launch: the GHz the kernel really sustained, and shader cycles per mixed
            # addition of one wavefront -- equal on two boxes that run the same instruction stream, whatever their clocks.
            "clock": ({"sclk_ghz_timed_loop": round(clk_loop["sclk_ghz"], 4), "sclk_ghz_single_step": round(clk_single["sclk_ghz"], 4) if clk_single else None,
                       "cycles_per_addition_timed_loop": round(clk_loop["cycles_per_addition"], 1),
                       "cycles_per_addition_single_step": round(clk_single["cycles_per_addition"], 1) if clk_single else None,
                       "k_accumulate_mcycles": round(acc_avg_ms * 1e-3 * clk_loop["sclk_ghz"] * 1e3, 3), "launches_sampled": int(clk_loop["samples"]),
                       "note": "k_accumulate_mcycles = avg_kernel_ms x sclk_ghz_timed_loop (10^6 shader cycles per launch of k_accumulate_pieces); cycles per "
                               "addition are those of the launch's first wavefront (its longest pieces), which shares its SIMD with two others"} if clk_loop and clk_loop["samples"] else None),
            "roofline": {"bound": "hbm", "kernel": "k_accumulate_pieces", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic, "traffic_source": traffic_src,
                         "traffic_over_algorithmic": round(traffic / alg_bytes, 3) if traffic else None,
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_kernel_ms": round(acc_avg_ms, 4), "launches_per_step": launches_per_step,
                         "launches_timed": int(acc_launches), "kernel_timing_every": max(1, args.kernel_timing_every),
                         "note": "avg_kernel_ms = hipEvents on the dispatch of every kernel_timing_every-th launch of the timed loop (the library's default times "
                                 "none: a timed dispatch costs the call ~11 us of launch overlap).  Integer-multiply (VALU) bound kernel; HBM fraction reported because BASELINE.json asks for it; "
                                 "the multiplier roofline is in roofline_valu.  avg_kernel_ms is k_accumulate_pieces alone: the work-item plan in "
                                 "front of it (k_place_count + k_piece_scatter: stage plan_ms) and k_combine_pieces behind it (combine_ms) are "
                                 "separate launches -- work the round-3 accumulation kernel did itself"},
            # the sort/scatter stages (north_star: "achieved HBM GB/s on the sort/scatter stages"): SURVEY section 8d algorithmic bytes
            # 8*N*W over the hipEvent time of k_coarse_hist .. k_fine_sort in the diagnostic step
            "roofline_sort": ({"bound": "hbm", "kernels": "k_coarse_hist+k_coarse_prefix+k_coarse_starts+k_coarse_scatter+k_fine_sort",
                               "algorithmic_bytes": sort_bytes, "ms": round(sort_ms, 4), "achieved": round(sort_bytes / (sort_ms * 1e-3) / 1e9, 1),
                               "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(sort_bytes / (sort_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5),
                               "traffic": sort_traffic, "traffic_source": sort_traffic_src,
                               "traffic_GBps": round(sort_traffic / (sort_ms * 1e-3) / 1e9, 1) if sort_traffic else None,
                               "note": "achieved = SURVEY's algorithmic bytes (one read + one write of the pairs) over the hipEvent time of the five launches; a "
                                       "counting sort reads the digits three times and writes them twice (traffic: what the counters saw), and three of the "
                                       "five launches are launch-bound (~5 us each); the placement of oversized regions (skewed scalars) runs in k_place_count, the plan stage"} if sort_ms > 0 else None),
            # the bound that actually holds: multiplier work of the launch against what two calibration
            # micro-kernels sustain on THIS device (dependent chains, 4 wavefronts per SIMD, like k_accumulate)
            "roofline_valu": valu_roofline(int(tm.get("num_adds", 0)), acc_avg_ms, mad_peak, fpmul_peak) if not args.streamed else None,
            "issue_floor": (issue_floor(int(tm.get("num_adds", 0)), round(acc_avg_ms * 1e-3 * clk_loop["sclk_ghz"] * 1e3, 3),
                                        4 * torch.cuda.get_device_properties(dev).multi_processor_count)
                            if clk_loop and clk_loop["samples"] and not args.streamed else None),
            "stage_ms_untimed_diagnostic_step": {k: round(v, 4) for k, v in tm.items() if k.endswith("_ms")},
        }
        if exchange is not None:
            out["exchange"] = exchange
        # projected strong scaling of THIS instance size from the per-shard single-GPU times committed under profiles/ (measured on
        # one device: what each rank would take alone) + the measured exchange of this run (or the committed estimate at N = 1)
        st_file = os.path.join(ROOT, "profiles", "shard_times.json")
        if os.path.exists(st_file):
            try:
                stj = json.load(open(st_file))
                tms = {int(k): float(v) for k, v in stj["device_call_ms_by_log2_points"].items()}
                # the exchange term: what THIS run measured when it ran on more than one rank; else an ESTIMATE -- the largest exchange this repo has
                # measured end to end (0.12 ms: two gloo ranks on one device, profiles/r4_final_bench_torchrun_2x_same_device.json), never less
                ex_meas = bool(exchange and world > 1)
                ex_est = exchange["ms_per_step"] if ex_meas else max(0.12, float(stj.get("exchange_ms_estimate", 0.12)))
                if args.log_n in tms:
                    proj = {}
                    for g in (2, 4, 8):
                        lg = args.log_n - g.bit_length() + 1
                        if lg in tms:
                            proj["x%d" % g] = {"ms": round(tms[lg] + ex_est, 4), "speedup": round(tms[args.log_n] / (tms[lg] + ex_est), 2)}
                    # the sizes the point-range sharding was designed for (BASELINE configs 4 and 5), next to this instance's
                    designed = {}
                    for big in (24, 26):
                        lg8 = big - 3
                        if lg8 in tms:
                            one = tms.get(big)
                            designed["2^%d_x8" % big] = {"ms": round(tms[lg8] + ex_est, 4), "shard_log2": lg8,
                                                         "speedup_vs_one_gpu": round(one / (tms[lg8] + ex_est), 2) if one else None}
                    out["projected_strong_scaling"] = {"from": "profiles/shard_times.json (single-GPU device calls, %s)" % stj.get("build", "?"),
                                                       "designed_for": designed,
                                                       "exchange_ms_used": round(ex_est, 4),
                                                       "exchange_ms_is": "measured in this run" if ex_meas else "ESTIMATE (no multi-GPU run: the largest exchange measured on one device)",
                                                       "one_gpu_ms": tms[args.log_n], **proj,
                                                       "note": "an estimate built on single-GPU shard times: no scaling curve has been measured on hardware"}
            except Exception:
                pass

        # ---- from here on: UNTIMED extras (host-pointer legs, resident / batch / table legs, the CPU baseline).  The measurement itself is
        #      complete; a watchdog prints the line without whatever is still missing if the extras stall (one run in ~80 on the gpurun pool
        #      sat for 20 minutes on a box that answered again afterwards -- cause unknown, never reproduced: 25 of 25 reruns took 8 s)
        import threading
        emit_lock = threading.Lock()  # the line is printed ONCE: by the watchdog (from a snapshot taken before the extras) or by the main thread
        emitted = [False]
        snapshot = json.dumps(dict(out, bit_exact=bit_exact,
                                   extras_timed_out="untimed legs did not finish within %d s; the timed measurement above is complete" % EXTRAS_DEADLINE_S))
        snapshot_ok = bit_exact

        def _emit_without_extras():
            with emit_lock:
                if emitted[0]:
                    return
                emitted[0] = True
                print(snapshot)
                sys.stdout.flush()
            os._exit(0 if snapshot_ok else 1)

        watchdog = threading.Timer(EXTRAS_DEADLINE_S, _emit_without_extras)
        watchdog.daemon = True
        if world == 1:
            watchdog.start()

        # ---- host-pointer legs (the reference's own measurement shape: benches/e2e.rs:46-60 times the call from HOST slices);
        #      same instance, copied to the host once outside every timed region; never `value`
        if world == 1 and not in_proc and not args.no_host_legs and not args.streamed:
            hb_t, hs_t = d_bases[0].cpu(), d_scalars[0].cpu()
            hb = hb_t.numpy().view(np.uint32).reshape(n_local, 16)
            hs = hs_t.numpy().view(np.uint32).reshape(n_local, 8)
            reps = 7 if n_local <= (1 << 22) else 3
            legs = {}
            r, avg, best = timed_calls(lambda: ctx.msm(hb, hs, mh.FORM_MONT), reps)
            legs["e2e_host_pageable_ms"], legs["e2e_host_pageable_min_ms"] = round(avg, 4), round(best, 4)
            legs["pageable_stream_chunks"] = ctx.timings()["stream_chunks"]
            ok = bool((r.affine_std == exp).all())
            hbp, hsp = hb_t.pin_memory(), hs_t.pin_memory()
            hbpn, hspn = hbp.numpy().view(np.uint32).reshape(n_local, 16), hsp.numpy().view(np.uint32).reshape(n_local, 8)
            r, avg, best = timed_calls(lambda: ctx.msm(hbpn, hspn, mh.FORM_MONT), reps)
            legs["e2e_host_pinned_ms"], legs["e2e_host_pinned_min_ms"] = round(avg, 4), round(best, 4)
            legs["pinned_stream_chunks"] = ctx.timings()["stream_chunks"]
            ok = ok and bool((r.affine_std == exp).all())
            # arkworks zero-copy: a [G1Affine] image (72-byte structs: x at 0, y at 32, `infinity` at 64) and Fr words taken as
            # Montgomery form, i.e. the scalars are s_i * R^-1 -- the expected point follows by linearity
            img = np.zeros((n_local, 72), np.uint8)
            img[:, :64] = hb.view(np.uint8).reshape(n_local, 64)
            r, avg, best = timed_calls(lambda: ctx.msm_arkworks(img, 72, 0, 32, 64, hs), reps)
            legs["e2e_arkworks_zero_copy_ms"], legs["e2e_arkworks_zero_copy_min_ms"] = round(avg, 4), round(best, 4)
            e2, e2i = expect(dot * pow(1 << 256, -1, R_ORDER))
            ok = ok and bool((r.affine_std == e2).all()) and r.is_infinity == bool(e2i)
            # prover shape: resident bases, a batch of host scalar vectors; single calls back to back vs
            # msm_bn254_g1_resident_batch (two MSMs in flight inside the context).  32 B per point of PCIe in both.
            ctx.upload_bases(hbpn, mh.FORM_MONT)
            K = 8
            r, avg, _ = timed_calls(lambda: [ctx.msm_resident(hspn) for _ in range(K)], 3)
            legs["resident_single_calls_ms_per_msm"] = round(avg / K, 4)
            ok = ok and all(bool((x.affine_std == exp).all()) for x in r)
            r, avg, _ = timed_calls(lambda: ctx.msm_resident_batch([hspn] * K), 3)
            legs["resident_batch_ms_per_msm"] = round(avg / K, 4)
            legs["resident_batch_layout"] = LAYOUT_NAMES.get(ctx.timings()["batch_layout"])  # AUTO: deterministic, by size
            ok = ok and all(bool((x.affine_std == exp).all()) for x in r)
            # ... and after the explicit, opt-in measurement of the three stream layouts on this context (msm_tune_batch)
            chosen, tune_ms = ctx.tune_batch([hspn] * K, reps=3)
            r, avg, _ = timed_calls(lambda: ctx.msm_resident_batch([hspn] * K), 3)
            legs["resident_batch_tuned_ms_per_msm"] = round(avg / K, 4)
            legs["resident_batch_tuned_layout"] = LAYOUT_NAMES.get(chosen)
            legs["resident_batch_tune_ms_per_msm"] = {LAYOUT_NAMES.get(k): round(v, 4) for k, v in tune_ms.items()}
            ok = ok and all(bool((x.affine_std == exp).all()) for x in r)
            # ... and with the scalars in HBM as well (a prover whose witness lives on the GPU): msm_bn254_g1_resident_device
            dsp = d_scalars[0].data_ptr()
            r, avg, _ = timed_calls(lambda: [ctx.msm_resident_device(dsp, n_local) for _ in range(K)], 3)
            legs["resident_device_scalars_ms_per_msm"] = round(avg / K, 4)
            ok = ok and all(bool((x.affine_std == exp).all()) for x in r)
            # row f4: the same resident set with its WINDOW TABLE (MSM_FLAG_WINDOW_TABLE: one bucket array shared by all windows)
            with mh.MsmContext(device=local_rank, flags=ctx_flags | mh.FLAG_WINDOW_TABLE) as tctx:
                tpl = mh.plan(n_local, args.window_bits, ctx_flags | mh.FLAG_WINDOW_TABLE)
                t_u = time.perf_counter()
                tctx.upload_bases(hbpn, mh.FORM_MONT)
                up_ms = (time.perf_counter() - t_u) * 1e3
                r, avg, _ = timed_calls(lambda: [tctx.msm_resident(hspn) for _ in range(K)], 3)
                legs["resident_table_single_calls_ms_per_msm"] = round(avg / K, 4)
                ok = ok and all(bool((x.affine_std == exp).all()) for x in r)
                r, avg, _ = timed_calls(lambda: tctx.msm_resident_batch([hspn] * K), 3)
                legs["resident_table_batch_ms_per_msm"] = round(avg / K, 4)
                ok = ok and all(bool((x.affine_std == exp).all()) for x in r)
                chosen, _tm = tctx.tune_batch([hspn] * K, reps=3)
                r, avg, _ = timed_calls(lambda: tctx.msm_resident_batch([hspn] * K), 3)
                legs["resident_table_batch_tuned_ms_per_msm"] = round(avg / K, 4)
                legs["resident_table_batch_tuned_layout"] = LAYOUT_NAMES.get(chosen)
                ok = ok and all(bool((x.affine_std == exp).all()) for x in r)
                r, avg, _ = timed_calls(lambda: [tctx.msm_resident_device(dsp, n_local) for _ in range(K)], 3)
                legs["resident_table_device_scalars_ms_per_msm"] = round(avg / K, 4)
                ok = ok and all(bool((x.affine_std == exp).all()) for x in r)
                legs["resident_table_plan"] = {"window_bits": tpl.window_bits, "num_windows": tpl.num_windows, "table_factor": tpl.table_factor,
                                               "bucket_arrays": tpl.bucket_arrays, "buckets_per_array": tpl.num_buckets, "glv_split": bool(tpl.glv),
                                               "table_MB": round(tpl.table_bytes / 1e6, 1), "upload_and_build_ms": round(up_ms, 1)}
            legs["bit_exact"] = ok
            legs["note"] = ("host-pointer calls on the same instance (PCIe-inclusive, 96-104 B per point); pageable = numpy arrays, "
                            "pinned = torch pin_memory; median and min of %d calls" % reps)
            out["host_pointer_legs"] = legs
            # the reference's OWN call shape (host slices in, benches/e2e.rs:46-60) beside `value` (inputs resident in HBM), never instead of it
            out["value_host_pinned_ms"] = legs["e2e_host_pinned_ms"]
            out["value_host_pageable_ms"] = legs["e2e_host_pageable_ms"]
            out["value_host_arkworks_ms"] = legs["e2e_arkworks_zero_copy_ms"]
            bit_exact = bit_exact and ok

        if not args.no_cpu_baseline and world == 1 and not in_proc:
            # CPU baseline: the arkworks-0.4-algorithm restatement (oracle_msm_pippenger) on this host's cores,
            # on the SAME bases/scalars (whole instance when it fits ~30 s of CPU work, else a prefix)
            threads = orc.threads_available()
            n_cpu = min(n_local, (1 << 20) if threads >= 8 else (1 << 18))
            if args.streamed:  # (the device copies were dropped: the instance lives in pinned host memory)
                hbc, hsc = hbn[:n_cpu], hsn[:n_cpu]
            else:
                hbc = d_bases[0][: n_cpu * 16].cpu().numpy().view(np.uint32).reshape(n_cpu, 16)
                hsc = d_scalars[0][: n_cpu * 8].cpu().numpy().view(np.uint32).reshape(n_cpu, 8)
            cpu_runs = []
            for _ in range(5):  # MEDIAN of five, spread in the line (a shared host: single runs moved by 35-43 % in rounds 2 and 3)
                t0 = time.perf_counter()
                cpu_aff, cpu_inf, _ = orc.msm_pippenger(hbc, hsc, orc.FORM_MONT, None, threads)
                cpu_runs.append((time.perf_counter() - t0) * 1e3)
            cpu_ms = sorted(cpu_runs)[len(cpu_runs) // 2]
            if n_cpu == n_local:
                cpu_ok = bool((cpu_aff == res.affine_std).all())
            else:
                chk = ctx.msm(hbc, hsc, mh.FORM_MONT) if args.streamed else ctx.msm_device(d_bases[0].data_ptr(), d_scalars[0].data_ptr(), n_cpu)
                cpu_ok = bool((cpu_aff == chk.affine_std).all())
            # arkworks parallelises over windows only (c = ln(n)*0.69 + 2 bits => 17 windows at 2^20): that many threads do work
            lg = int(np.log2(n_cpu))
            c_ark = 3 if n_cpu < 32 else (lg * 69) // 100 + 2
            busy = min(threads, -(-254 // c_ark))
            out["cpu_baseline"] = {"value": round(cpu_ms, 2), "all_runs_ms": [round(x, 1) for x in cpu_runs],
                                   "spread": round((max(cpu_runs) - min(cpu_runs)) / cpu_ms, 3), "unit": "ms", "cores": busy, "kind": "port",
                                   "sample": "first 2^%d points of the same instance, one MSM, MEDIAN of five runs (spread = (max - min) / median), threads "
                                             "pinned (OMP_PROC_BIND=%s OMP_PLACES=%s); arkworks-0.4 algorithm restated in C "
                                             "(not arkworks itself): one thread per window, %d windows of %d bits, %d host threads available"
                                             % (lg, os.environ.get("OMP_PROC_BIND"), os.environ.get("OMP_PLACES"), -(-254 // c_ark), c_ark, threads),
                                   "agrees_with_gpu": cpu_ok}
            bit_exact = bit_exact and cpu_ok
        watchdog.cancel()
        out["bit_exact"] = bit_exact
        with emit_lock:
            if not emitted[0]:
                emitted[0] = True
                print(json.dumps(out))
                sys.stdout.flush()
    if ctx is not None:
        ctx.close()
    if multi is not None:
        multi.close()
    if world > 1:
        ok_t = torch.tensor([1 if bit_exact else 0], dtype=torch.int32, device=xdev if xdev is not None else "cpu")
        dist.broadcast(ok_t, src=0)
        bit_exact = bool(ok_t.item())
        dist.barrier()
        dist.destroy_process_group()
    if not bit_exact:
        sys.exit(1)  # a wrong answer must not look like a timing


if __name__ == "__main__":
    main()
