"""CPU, world_size 2, gloo: the N>1 path of bench.py -- contiguous sharding of the batch, no
data-path collective, one gather of results -- gives exactly the single-process answer.  The
per-rank solve is played by the CPU oracle here (no GPU in this container); on the GPU box the
same sharding.py functions wrap the HIP solver with backend "nccl" (= RCCL)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tape, cls, total, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tenscalc_amd.problems.sls import sls_batch
    from tenscalc_amd.sharding import gather_results, pack_results, shard_range, unpack_results
    from oracle.tc_oracle import Oracle
    lo, hi = shard_range(total, rank, world)
    A, b, x0 = sls_batch(hi - lo, seed0=300 + lo)
    o = Oracle(tape)
    out, st, it = o.solve_batch({cls + "_set_A": np.transpose(A, (0, 2, 1)).reshape(hi - lo, -1),
                                 cls + "_set_b": b, cls + "_set_x": x0}, cls + "_getOutputs", hi - lo,
                                mu0=1.0, maxIter=30)
    packed = pack_results(torch.from_numpy(out.T.copy()), torch.from_numpy(st), torch.from_numpy(it))
    g = gather_results(packed, world)
    O, ST, IT = unpack_results(g, out.shape[1])
    dist.barrier()
    if rank == 0:
        q.put((O.numpy(), ST.numpy(), IT.numpy()))
    dist.destroy_process_group()


def test_two_rank_sharding_matches_single_process(sls_spec, sls_tape_path):
    from tenscalc_amd.problems.sls import sls_batch
    from oracle.tc_oracle import Oracle
    total, world = 24, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, sls_tape_path, sls_spec.classname, total, q))
             for r in range(world)]
    for p in procs:
        p.start()
    O, ST, IT = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    A, b, x0 = sls_batch(total, seed0=300)
    c = sls_spec.classname
    ref, rst, rit = Oracle(sls_tape_path).solve_batch(
        {c + "_set_A": np.transpose(A, (0, 2, 1)).reshape(total, -1), c + "_set_b": b, c + "_set_x": x0},
        c + "_getOutputs", total, mu0=1.0, maxIter=30)
    assert np.array_equal(O.T, ref) and np.array_equal(ST, rst) and np.array_equal(IT, rit)


def test_shard_range_partitions_everything():
    from tenscalc_amd.sharding import shard_range
    for total in (0, 1, 7, 64, 65536, 65537):
        for world in (1, 2, 3, 8):
            r = [shard_range(total, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == total
            assert all(r[k][1] == r[k + 1][0] for k in range(world - 1))
            sizes = [hi - lo for lo, hi in r]
            assert max(sizes) - min(sizes) <= 1


def _bench_flow_worker(rank, world, port, every, steps, q):
    """bench.py's N>1 control flow (tenscalc_amd/sharding.py: run_timed + ResultExchange + reduce_timing) with the
    per-rank solve replaced by a stand-in that produces recognisable results"""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tenscalc_amd.sharding import ResultExchange, reduce_timing, run_timed
    B, R = 5, 4
    state = dict(k=-1, calls=0)

    def step(k):
        state["k"] = k
        state["calls"] += 1
        if rank == 1:
            import time
            time.sleep(.01)                      # the slower rank sets the reported time

    def results():
        return torch.full((R, B), float(100 * rank + state["k"]), dtype=torch.float64)
    xch = ResultExchange(world, (R, B), torch.device("cpu"))
    seen = []
    orig = xch.exchange

    def counting(p):
        seen.append(float(p[0, 0]))
        orig(p)
    xch.exchange = counting
    el = run_timed(steps, step, results, world, every=every, barrier=dist.barrier, exchange=xch)
    elapsed, ok = reduce_timing(el, 3 + rank, world, torch.device("cpu"))
    q.put((rank, state["calls"], seen, xch.last().clone().numpy(), el, elapsed, ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("every", [False, True])
def test_bench_multi_rank_control_flow(every):
    """world_size 2 over gloo: exactly K steps per rank; ONE gather after the K steps by default, one per
    step (double-buffered, drained before the clock stops) with --gather-every-step; every rank ends with everybody's
    last results; the reported time is the MAX over ranks and the solved count the SUM"""
    world, steps = 2, 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_flow_worker, args=(r, world, port, every, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, calls, seen, last, el, elapsed, ok in res:
        assert calls == steps
        assert seen == ([100.0 * rank + k for k in range(steps)] if every else [100.0 * rank + steps - 1])
        assert last.shape == (world, 4, 5)
        assert (last[0] == steps - 1).all() and (last[1] == 100 + steps - 1).all()      # everybody's LAST results
        assert ok == 3 + 4
    assert res[0][5] == res[1][5] == max(res[0][4], res[1][4])                            # max over ranks
    assert res[1][4] >= steps * .01
