"""Multi-GPU decomposition of one MSM: one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm).

MSM is linear, so the instance shards by contiguous POINT RANGE (no data-path collective while the GPUs work);
the only exchange is the per-rank partial group element.  EC addition is not an RCCL reduction op, so the
"all-reduce of partial sums" is an all-gather of one 96-byte Jacobian point per rank followed by a local fold
in rank order -- identical bits on every rank.  The reference has no multi-device code at all.
"""
import numpy as np

from . import ERR_HIP, FLAG_DETERMINISTIC, OK, MsmError, MsmResult, combine_partials

WORDS = 25  # what a rank sends: 24 words of its partial (Jacobian, Montgomery) + 1 status word (C-ABI code, 0 = ok)


def shard_range(n_total, rank, world):
    """contiguous point range [lo, hi) of `rank`"""
    return rank * n_total // world, (rank + 1) * n_total // world


class _Exchange:
    """Reusable buffers of the 96-byte exchange: pinned host staging on both sides and device send/receive tensors, so a
    step costs two asynchronous 96 B / 96*world B copies, one RCCL all-gather and one stream synchronisation -- no
    allocation, no pageable staging copy."""

    def __init__(self, world, device):
        import torch

        self.world, self.device = world, device
        on_gpu = device is not None and torch.device(device).type == "cuda"
        self.h_send = torch.empty(WORDS, dtype=torch.int32, pin_memory=on_gpu)
        self.h_recv = torch.empty(WORDS * world, dtype=torch.int32, pin_memory=on_gpu)
        self.d_send = torch.empty(WORDS, dtype=torch.int32, device=device) if on_gpu else None
        self.d_recv = torch.empty(WORDS * world, dtype=torch.int32, device=device) if on_gpu else None
        self._send_np = self.h_send.numpy()
        self._recv_np = self.h_recv.numpy().reshape(world, WORDS)
        self.last_ms = 0.0  # wall clock of the last exchange on this rank (bench.py reports it)

    def run(self, partial, status, group):
        """-> (partials (world, 24) uint32, status (world,) int32), the same on every rank"""
        import time

        import torch
        import torch.distributed as dist

        t0 = time.perf_counter()
        self._send_np[:24] = np.ascontiguousarray(partial, dtype=np.uint32).view(np.int32).reshape(24)
        self._send_np[24] = status
        if self.d_send is None:  # CPU group (gloo): the tests' path
            dist.all_gather_into_tensor(self.h_recv, self.h_send, group=group)
        else:
            self.d_send.copy_(self.h_send, non_blocking=True)
            dist.all_gather_into_tensor(self.d_recv, self.d_send, group=group)
            self.h_recv.copy_(self.d_recv, non_blocking=True)
            torch.cuda.current_stream(self.device).synchronize()
        self.last_ms = (time.perf_counter() - t0) * 1e3
        return self._recv_np[:, :24].copy().view(np.uint32), self._recv_np[:, 24].copy()


_exchanges = {}


def _exchange(device, group):
    import torch.distributed as dist

    world = dist.get_world_size(group)
    key = (world, str(device), id(group))
    ex = _exchanges.get(key)
    if ex is None:
        ex = _exchanges[key] = _Exchange(world, device)
    return ex


def last_exchange_ms(device=None, group=None):
    """wall clock of this rank's last all-gather + copies (0.0 before the first one)"""
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return 0.0
    return _exchange(device, group).last_ms


def all_gather_partials(partial_jacobian_mont, device=None, group=None):
    """all-gather of the 24-word partial of every rank -> (world, 24) uint32 array, same on all ranks"""
    return _exchange(device, group).run(partial_jacobian_mont, OK, group)[0]


def all_reduce_msm(local_result, device=None, group=None, flags=0) -> MsmResult:
    """exchange + fold: every rank ends with the full MSM result.

    flags: the contexts' MSM_FLAG_DETERMINISTIC, if they carry it -- the fold then hands out the canonical Z = 1 Jacobian words, the same 24
    words a single GPU or msm_multi returns for this group element (msm_bn254_g1_combine_flags).

    `local_result` is this rank's MsmResult -- or the MsmError its local MSM raised.  NO RANK MAY HANG: a rank whose local MSM
    failed (one scalar >= 2^254 in its shard is enough) still joins the all-gather, with the identity as its partial and its
    status code in the 25th word; every rank then raises MsmError with the FIRST failing rank's code (the reference returns Err,
    metal_msm.rs:647-656 -- it never leaves a caller waiting)."""
    import torch.distributed as dist

    failed = isinstance(local_result, MsmError)
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        if failed:
            raise local_result
        return local_result
    partial = np.zeros(24, np.uint32) if failed else local_result.jacobian_mont  # Z = 0: the identity
    parts, status = _exchange(device, group).run(partial, int(local_result.code) if failed else OK, group)
    bad = np.nonzero(status)[0]
    if bad.size:
        r = int(bad[0])
        own = " (this rank: %s)" % local_result if failed else ""
        raise MsmError(int(status[r]), "rank %d of %d failed with status %d%s" % (r, len(status), int(status[r]), own))
    return combine_partials(parts, want_affine=False, flags=flags & FLAG_DETERMINISTIC)


def guarded(local_call, device=None, group=None, flags=0) -> MsmResult:
    """run this rank's local MSM (`local_call()` -> MsmResult) and all-reduce it; an MsmError raised by the local call travels
    through the exchange instead of keeping this rank out of the collective -- and so does ANY other exception (status ERR_HIP on
    the peers; this rank re-raises the original exception after the exchange)"""
    pending = None
    try:
        local = local_call()
    except MsmError as e:
        local = e
    except Exception as e:  # argument packing (ValueError), torch / HIP RuntimeError (out of memory) ...: the peers are already on
        pending = e         # their way into the all-gather, so this rank joins it too, with a generic status, and re-raises afterwards
        local = MsmError(ERR_HIP, "%s: %s" % (type(e).__name__, e))
    try:
        return all_reduce_msm(local, device, group, flags)
    except MsmError:
        if pending is not None:
            raise pending
        raise


def distributed_msm_device(ctx, d_bases_ptr, d_scalars_ptr, n_local, device=None, group=None, d_inf_ptr=None) -> MsmResult:
    """this rank's shard is already in its HBM: run the HIP pipeline on it, then all-reduce the partials"""
    return guarded(lambda: ctx.msm_device(d_bases_ptr, d_scalars_ptr, n_local, d_inf_ptr), device, group, getattr(ctx, "flags", 0))
