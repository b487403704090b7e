"""Multi-GPU decomposition of one MSM: one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm).

MSM is linear, so the instance shards by contiguous POINT RANGE (no data-path collective while the GPUs work);
the only exchange is the per-rank partial group element.  EC addition is not an RCCL reduction op, so the
"all-reduce of partial sums" is an all-gather of one 96-byte Jacobian point per rank followed by a local fold
in rank order -- identical bits on every rank.  The reference has no multi-device code at all.
"""
import numpy as np

from . import MsmResult, combine_partials


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
        self.h_send = torch.empty(24, dtype=torch.int32, pin_memory=on_gpu)
        self.h_recv = torch.empty(24 * world, dtype=torch.int32, pin_memory=on_gpu)
        self.d_send = torch.empty(24, dtype=torch.int32, device=device) if on_gpu else None
        self.d_recv = torch.empty(24 * world, dtype=torch.int32, device=device) if on_gpu else None
        self._send_np = self.h_send.numpy()
        self._recv_np = self.h_recv.numpy().view(np.uint32).reshape(world, 24)

    def run(self, partial, group):
        import torch
        import torch.distributed as dist

        self._send_np[:] = np.ascontiguousarray(partial, dtype=np.uint32).view(np.int32).reshape(24)
        if self.d_send is None:  # CPU group (gloo): the tests' path
            dist.all_gather_into_tensor(self.h_recv, self.h_send, group=group)
        else:
            self.d_send.copy_(self.h_send, non_blocking=True)
            dist.all_gather_into_tensor(self.d_recv, self.d_send, group=group)
            self.h_recv.copy_(self.d_recv, non_blocking=True)
            torch.cuda.current_stream(self.device).synchronize()
        return self._recv_np.copy()


_exchanges = {}


def all_gather_partials(partial_jacobian_mont, device=None, group=None):
    """all-gather of the 24-word partial of every rank -> (world, 24) uint32 array, same on all ranks"""
    import torch.distributed as dist

    world = dist.get_world_size(group)
    key = (world, str(device), id(group))
    ex = _exchanges.get(key)
    if ex is None:
        ex = _exchanges[key] = _Exchange(world, device)
    return ex.run(partial_jacobian_mont, group)


def all_reduce_msm(local_result: MsmResult, device=None, group=None) -> MsmResult:
    """exchange + fold: every rank ends with the full MSM result"""
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local_result
    return combine_partials(all_gather_partials(local_result.jacobian_mont, device, group), want_affine=False)


def distributed_msm_device(ctx, d_bases_ptr, d_scalars_ptr, n_local, device=None, group=None, d_inf_ptr=None) -> MsmResult:
    """this rank's shard is already in its HBM: run the HIP pipeline on it, then all-reduce the partials"""
    return all_reduce_msm(ctx.msm_device(d_bases_ptr, d_scalars_ptr, n_local, d_inf_ptr), device, group)
