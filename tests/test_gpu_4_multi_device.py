"""GPU tests that ARM THEMSELVES on a box with two or more visible MI355X (VERDICT r3 "missing" 1-2): on today's 1-GPU boxes every test
here is skipped; on the first multi-GPU box they exercise what no hardware has run yet --

  * msm_multi on ALL visible devices under BOTH exchanges: ncclCommInitAll over > 1 device + ncclAllGather of the 96-byte partials over xGMI, and
    the host fold; one context + host thread per device, the fold in rank order; what each exchange cost is written to
    gpurun_out/multi_exchange_times.json, and AUTO must report the probe it ran at creation and keep the faster one (csrc/msm_multi.inc);
  * the same with a failing rank (every caller gets MSM_ERR_BAD_ARG, nobody waits in the collective) and with MSM_HIP_MULTI_VERIFY=1;
  * BASELINE config 4 at full size (2^24 points over the visible devices) and the headline size 2^20, by the closed form;
  * the one-process-per-GPU path that bench.py --gpus N runs: torchrun with nproc = device_count on the "nccl" (= RCCL) backend.

The reference has no multi-device code (host/gpu.rs:3-5 opens the system default device); the partitioning is BASELINE.json's
("point-range shard + RCCL partial-sum all-reduce")."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import mopro_msm_hip as mh
from mopro_msm_hip import testhooks as th
from conftest import ROOT
from oracle import bn254_oracle as orc

pytestmark = pytest.mark.gpu
R = orc.R_ORDER
MUL, MASK = 0xD1342543DE82EF95, (1 << 64) - 1


def _ndev():
    try:
        import torch
        return torch.cuda.device_count()  # (counting devices does not initialise the GPU on this image)
    except Exception:
        return 0


needs_two = pytest.mark.skipif(_ndev() < 2, reason="needs >= 2 visible GPUs (arms itself on a multi-GPU box)")


def _ints(words):
    a = np.ascontiguousarray(words, dtype=np.uint32).reshape(-1, 8)
    cols = [a[:, j].tolist() for j in range(8)]
    out = []
    for i in range(a.shape[0]):
        v = 0
        for j in range(7, -1, -1):
            v = (v << 32) | cols[j][i]
        out.append(v)
    return out


def _expected(dot):
    g = np.zeros(16, np.uint32)
    g[0], g[8] = 1, 2
    return orc.g1_to_affine_std(orc.g1_scalar_mul(g, orc.int_to_words(dot % R)))


class Sharded:
    """one instance of 2^logn points cut into point-range shards, shard g generated ON device g (bases k_i*G, scalars s_i); the closed
    form's dot product is accumulated shard by shard on the host (chunks of 2^20 logs: a 2^24 instance never holds all of them)"""

    def __init__(self, logn, G, seed=0xB2540A00):
        import torch
        self.n, self.G = 1 << logn, G
        self.cuts = [g * self.n // G for g in range(G + 1)]
        self.d_b, self.d_s, self.dot = [], [], 0
        for g in range(G):
            lo, hi = self.cuts[g], self.cuts[g + 1]
            dev = torch.device("cuda", g)
            b = torch.empty((hi - lo) * 16, dtype=torch.int32, device=dev)
            s = torch.empty((hi - lo) * 8, dtype=torch.int32, device=dev)
            with th.HooksContext(device=g) as gen:
                gen.generate_device((seed + lo * MUL) & MASK, (seed + 1 + lo * MUL) & MASK, hi - lo, b.data_ptr(), s.data_ptr())
            torch.cuda.synchronize(dev)
            self.d_b.append(b)
            self.d_s.append(s)
            for c0 in range(lo, hi, 1 << 20):
                cnt = min(1 << 20, hi - c0)
                k = _ints(th.generate_scalars_host((seed + c0 * MUL) & MASK, cnt, nonzero=True))
                sc = _ints(th.generate_scalars_host((seed + 1 + c0 * MUL) & MASK, cnt))
                self.dot += sum(a * b_ for a, b_ in zip(k, sc))
        self.exp, _ = _expected(self.dot)

    def call(self, m, scalars=None):
        s = scalars or self.d_s
        return m.msm_device([t.data_ptr() for t in self.d_b], [t.data_ptr() for t in s], [self.cuts[g + 1] - self.cuts[g] for g in range(self.G)])


@needs_two
@pytest.mark.parametrize("logn", [20, 24])
def test_msm_multi_on_all_visible_devices_both_exchanges(logn):
    """MsmMulti on every visible device under BOTH exchanges -- explicit RCCL (ncclCommInitAll + ncclAllGather of 24 words per rank) and
    the host fold: every device gets its point range, the result is the closed form at 2^20 (the headline size) and 2^24 (BASELINE
    config 4), the affine words of the two exchanges are identical, and what each exchange cost is RECORDED
    (gpurun_out/multi_exchange_times.json: the first multi-GPU run decides the default instead of assuming it -- VERDICT r4 item 4).
    AUTO measures both exchanges when the handle is created and keeps the faster one: it must report what it measured and which it kept."""
    import json
    import time
    G = _ndev()
    inst = Sharded(logn, G)
    rec = {"devices": G, "log_n": logn}
    outs = {}
    for name, mode in (("rccl", mh.EXCHANGE_RCCL), ("host", mh.EXCHANGE_HOST)):
        with mh.MsmMulti(exchange=mode) as m:
            assert m.num_devices == G and m.exchange == mode
            assert m.exchange_probe() == (0.0, 0.0), "an explicit exchange is never probed"
            walls, exs = [], []
            for it in range(6):
                t0 = time.perf_counter()
                r = inst.call(m)
                walls.append((time.perf_counter() - t0) * 1e3)
                assert not r.is_infinity and (r.affine_std == inst.exp).all()
                ex_ms, shard_ms = m.exchange_stats()
                assert len(shard_ms) == G and all(t > 0 for t in shard_ms) and ex_ms > 0
                exs.append(ex_ms)
            assert sum(m.timings(g)["num_points"] for g in range(G)) == inst.n
            outs[name] = r.affine_std.copy()
            rec[name] = {"call_ms_min": round(min(walls[1:]), 4), "call_ms_median": round(sorted(walls[1:])[2], 4),
                         "exchange_ms_min": round(min(exs[1:]), 4), "shard_ms_last": [round(t, 4) for t in shard_ms]}
    assert (outs["rccl"] == outs["host"]).all()
    with mh.MsmMulti() as m:  # AUTO
        p_rccl, p_host = m.exchange_probe()
        assert p_rccl > 0 and p_host > 0, "AUTO on distinct devices with librccl must have measured both exchanges"
        assert m.exchange == (mh.EXCHANGE_RCCL if p_rccl < p_host else mh.EXCHANGE_HOST), "AUTO must keep the exchange it measured faster"
        assert (inst.call(m).affine_std == inst.exp).all()
        rec["auto"] = {"probe_rccl_ms": round(p_rccl, 4), "probe_host_ms": round(p_host, 4), "kept": "rccl" if m.exchange == mh.EXCHANGE_RCCL else "host"}
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(out_dir, "multi_exchange_times.json")
    try:
        allrec = json.load(open(path))
    except Exception:
        allrec = []
    allrec.append(rec)
    json.dump(allrec, open(path, "w"), indent=1)
    print("multi-GPU exchange times:", json.dumps(rec))


@needs_two
def test_msm_multi_host_pointers_each_device_pulls_its_own_range():
    """the drop-in call on host pointers through msm_multi: every device streams ITS point range over its own PCIe link"""
    import torch
    G = _ndev()
    inst = Sharded(20, G, seed=0xB2540B00)
    hb = np.concatenate([t.cpu().numpy().view(np.uint32).reshape(-1, 16) for t in inst.d_b])
    hs = np.concatenate([t.cpu().numpy().view(np.uint32).reshape(-1, 8) for t in inst.d_s])
    with mh.MsmMulti(exchange=mh.EXCHANGE_RCCL) as m:
        assert m.exchange == mh.EXCHANGE_RCCL
        r = m.msm(hb, hs, mh.FORM_MONT)
        assert (r.affine_std == inst.exp).all()
    with mh.MsmMulti(exchange=mh.EXCHANGE_HOST) as m:  # the host fold gives the same bits
        assert (m.msm(hb, hs, mh.FORM_MONT).affine_std == inst.exp).all()
    del torch


@needs_two
def test_msm_multi_bad_scalar_in_one_shard_fails_every_rank_and_the_handle_survives():
    """a scalar >= 2^254 in the LAST device's shard: the ranks rendezvous on the host in front of the collective and all skip it, the call
    returns MSM_ERR_BAD_ARG naming the rank -- nobody waits in ncclAllGather -- and the next, clean call on the same handle is right"""
    import torch
    G = _ndev()
    inst = Sharded(20, G, seed=0xB2540C00)
    bad = [t for t in inst.d_s]
    bad[G - 1] = inst.d_s[G - 1].clone()
    bad[G - 1][8 * 12345 + 7] = 0x40000000
    torch.cuda.synchronize(torch.device("cuda", G - 1))
    with mh.MsmMulti(exchange=mh.EXCHANGE_RCCL) as m:
        assert m.exchange == mh.EXCHANGE_RCCL
        with pytest.raises(mh.MsmError) as e:
            inst.call(m, bad)
        assert e.value.code == mh.ERR_BAD_ARG and ("rank %d" % (G - 1)) in str(e.value), str(e.value)
        assert (inst.call(m).affine_std == inst.exp).all()


@needs_two
def test_msm_multi_verify_all_ranks_hold_the_same_bits(monkeypatch):
    """MSM_HIP_MULTI_VERIFY=1 at creation (a debug knob of the HOOKS build): after the all-gather every rank's folded 24 words are compared"""
    monkeypatch.setenv("MSM_HIP_MULTI_VERIFY", "1")
    G = _ndev()
    inst = Sharded(18, G, seed=0xB2540D00)
    with mh.MsmMulti(exchange=mh.EXCHANGE_RCCL, _lib=th.load_hooks_library()) as m:
        assert m.exchange == mh.EXCHANGE_RCCL
        assert (inst.call(m).affine_std == inst.exp).all()


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _torchrun_all_devices(extra_env=None):
    G = _ndev()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MSM_TEST_BACKEND="nccl", HSA_ENABLE_IPC_MODE_LEGACY="0", **(extra_env or {}))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(G), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_gpu_worker.py")]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    return p, [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]


@needs_two
def test_one_process_per_gpu_over_rccl():
    """what bench.py --gpus N runs: torchrun, one rank per DEVICE, backend "nccl" (RCCL): every rank ends with the same, correct bits"""
    G = _ndev()
    p, lines = _torchrun_all_devices({"MSM_TEST_LOG_N": "20"})
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert len(lines) == G and all(l["ok"] for l in lines), lines
    assert len({json.dumps(l["affine"]) for l in lines}) == 1
    assert sorted(l["device"] for l in lines) == list(range(G)) and all(l["backend"] == "nccl" for l in lines)


@needs_two
def test_one_process_per_gpu_failing_rank_over_rccl():
    """a scalar >= 2^254 in rank 1's shard: it joins the RCCL all-gather with the identity and its status word, every rank raises
    MsmError(ERR_BAD_ARG) naming rank 1 -- none hangs (metal_msm.rs:647-656 returns Err)"""
    G = _ndev()
    p, lines = _torchrun_all_devices({"MSM_TEST_BAD_SCALAR_RANK": "1"})
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert len(lines) == G and all(l["ok"] and l["error_code"] == mh.ERR_BAD_ARG for l in lines), lines


@needs_two
def test_bench_gpus_n_starts_plainly():
    """`python bench.py --gpus N` with NO torchrun around it (how the driver starts the N = 1 line) must launch its ranks itself"""
    G = _ndev()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(G), "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=1200, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == G and line["bit_exact"] and line["exchange"]["world_seen"] == G
    assert line["exchange"]["devices_seen"] == list(range(G))
