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


def all_gather_partials(partial_jacobian_mont, device=None, group=None):
    """all-gather of the 24-word partial of every rank -> (world, 24) uint32 array, same on all ranks"""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    mine = torch.from_numpy(np.ascontiguousarray(partial_jacobian_mont, dtype=np.uint32).view(np.int32).copy())
    if device is not None:
        mine = mine.to(device)
    out = torch.empty(24 * world, dtype=torch.int32, device=mine.device)
    dist.all_gather_into_tensor(out, mine, group=group)
    return out.cpu().numpy().view(np.uint32).reshape(world, 24)


def all_reduce_msm(local_result: MsmResult, device=None, group=None) -> MsmResult:
    """exchange + fold: every rank ends with the full MSM result"""
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local_result
    return combine_partials(all_gather_partials(local_result.jacobian_mont, device, group))


def distributed_msm_device(ctx, d_bases_ptr, d_scalars_ptr, n_local, device=None, group=None, d_inf_ptr=None) -> MsmResult:
    """this rank's shard is already in its HBM: run the HIP pipeline on it, then all-reduce the partials"""
    return all_reduce_msm(ctx.msm_device(d_bases_ptr, d_scalars_ptr, n_local, d_inf_ptr), device, group)
