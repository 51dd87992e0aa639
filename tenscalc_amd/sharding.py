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


class ResultExchange:
    """The path's only exchange, as bench.py runs it for N > 1: every rank contributes its packed results
    ([R, B] float64) and receives everybody's.  Double buffered and asynchronous: a gather is enqueued right after
    the step that produced the results and is only waited for when its buffer comes up for reuse -- or, for all of
    them, by drain() before the timed region ends."""

    def __init__(self, world, shape, device, dtype=None):
        import torch
        self.world = int(world)
        self.gathered = [torch.empty((self.world,) + tuple(shape), dtype=dtype or torch.float64, device=device)
                         for _ in range(2)]
        self.packed = [None, None]
        self.pending = [None, None]
        self.count = 0
        self.device = device

    def exchange(self, packed):
        import torch.distributed as dist
        k = self.count & 1
        self.count += 1
        if self.pending[k] is not None:
            self.pending[k].wait()
        self.packed[k] = packed.to(self.device).contiguous()       # kept alive until the collective has run
        g = self.gathered[k]
        self.pending[k] = dist.all_gather_into_tensor(g.view(self.world * g.shape[1], g.shape[2]), self.packed[k],
                                                      async_op=True)

    def drain(self):
        for k in range(2):
            if self.pending[k] is not None:
                self.pending[k].wait()
                self.pending[k] = None

    def last(self):
        """[world, R, B] of the most recent exchange (after drain())"""
        return self.gathered[(self.count - 1) & 1] if self.count else None


def run_timed(steps, step, results, world, every=False, barrier=None, sync=None, exchange=None):
    """bench.py's timed region: barrier + sync, exactly `steps` calls of step(k), the result gather(s) -- after every
    step (`every`) or ONE after the last step -- drained, sync + barrier; returns the local elapsed
    seconds.  results() -> the rank's packed results tensor; exchange: a ResultExchange (world > 1)."""
    import time
    barrier = barrier or (lambda: None)
    sync = sync or (lambda: None)
    if world > 1:
        barrier()
    sync()
    t0 = time.perf_counter()
    for k in range(steps):
        step(k)
        if world > 1 and every:
            exchange.exchange(results())
    if world > 1:
        if not every:
            exchange.exchange(results())
        exchange.drain()
    sync()
    if world > 1:
        barrier()
        sync()
    return time.perf_counter() - t0


def reduce_timing(elapsed, solved, world, device):
    """(max elapsed over ranks, total solved instances over ranks): what rank 0 reports"""
    import torch
    import torch.distributed as dist
    el = torch.tensor([float(elapsed)], dtype=torch.float64, device=device)
    ok = torch.tensor([float(solved)], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(ok, op=dist.ReduceOp.SUM)
    return float(el.item()), float(ok.item())
