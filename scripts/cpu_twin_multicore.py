"""The CPU baseline on several host cores: P worker PROCESSES (the reference's generated C keeps its state in
globals, so it is one instance per process, SURVEY.md 8b "Threading"), each with its own copy of the sls CPU
twin (oracle/c_emitter.py + reference lib/ipmPD_CSsolver.c, gcc -O1), solving its slice of the same batch.
No GPU is touched.    python scripts/cpu_twin_multicore.py [P=16] [instances per worker=20000]"""
import multiprocessing as mp
import os
import sys
import time

sys.path.insert(0, os.getcwd())


def worker(rank, n, q):
    import numpy as np
    import __graft_entry__ as g
    from oracle.c_emitter import CpuTwin
    from tenscalc_amd.problems.sls import SOLVE_ARGS
    spec = g.sls_spec("strict")
    tw = CpuTwin(spec)
    rs = np.random.RandomState(100 + rank)
    A = rs.random_sample((n, 800)); b = rs.random_sample((n, 100)); x0 = .02 * rs.random_sample((n, 8))
    tw.solve_batch({"A": A[:64], "b": b[:64], "x": x0[:64]}, 64, mu0=SOLVE_ARGS["mu0"], maxIter=SOLVE_ARGS["maxIter"])
    q.put(("ready", rank))
    t0 = time.perf_counter()
    out, st, it = tw.solve_batch({"A": A, "b": b, "x": x0}, n, mu0=SOLVE_ARGS["mu0"], maxIter=SOLVE_ARGS["maxIter"])
    q.put(("done", rank, time.perf_counter() - t0, int((st == 0).sum())))


if __name__ == "__main__":
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, n, q)) for r in range(P)]
    t0 = time.perf_counter()
    for p in ps:
        p.start()
    res = []
    while len(res) < P:
        m = q.get(timeout=600)
        if m[0] == "done":
            res.append(m)
    for p in ps:
        p.join()
    wall = max(r[2] for r in res)
    ok = sum(r[3] for r in res)
    print(f"sls CPU twin on {P} processes ({os.cpu_count()} host cores visible): {ok} solves, slowest worker {wall:.2f} s "
          f"-> {ok / wall:.3e} solves/s aggregate, {ok / wall / P:.3e} per process")
