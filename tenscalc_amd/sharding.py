"""Multi-GPU sharding of a batch of independent problem instances.

The reference has no distributed code at all (one instance per process).  The batched
path shards embarrassingly: rank r of W owns the contiguous instance range
``[r*B/W, (r+1)*B/W)`` on its own GPU, nothing is exchanged during the solve, and the only
collective is the gather of results (outputs + status + iteration counts), one
``all_gather_into_tensor`` over RCCL/xGMI (backend "nccl") -- or gloo in the CPU tests.
"""
from __future__ import annotations


def shard_range(total, rank, world):
    """contiguous [lo, hi) of `total` instances owned by `rank` (sizes differ by at most 1)"""
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pack_results(outputs, status, iters):
    """[nOut, B] float64 outputs + int32 status/iter -> one [nOut+2, B] float64 tensor (exact:
    the int32 values are far below 2**53)"""
    import torch
    return torch.cat([outputs, status.to(torch.float64)[None], iters.to(torch.float64)[None]], 0)


def gather_results(packed, world, out=None):
    """all ranks contribute an equally shaped [R, B] tensor; returns [world, R, B] on every rank"""
    import torch
    import torch.distributed as dist
    if world == 1:
        return packed[None]
    if out is None:
        out = torch.empty((world,) + tuple(packed.shape), dtype=packed.dtype, device=packed.device)
    dist.all_gather_into_tensor(out.view(world * packed.shape[0], packed.shape[1]), packed.contiguous())
    return out


def unpack_results(gathered, n_out):
    """[world, nOut+2, B] -> (outputs [nOut, world*B], status [world*B] int32, iters [world*B] int32),
    instance order = rank-major, i.e. the global contiguous order of shard_range with equal shards"""
    import torch
    W, R, B = gathered.shape
    g = gathered.permute(1, 0, 2).reshape(R, W * B)
    return g[:n_out], g[n_out].to(torch.int32), g[n_out + 1].to(torch.int32)
