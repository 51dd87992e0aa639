"""Measurement aid (GPU box): sls batches in flight on N handles / N streams -- how much of a launch's tail the next
launch can fill.  python scripts/sls_two_streams.py [variant] [N ...]"""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import torch

import __graft_entry__ as g
from tenscalc_amd.problems.sls import SOLVE_ARGS
from tenscalc_amd.runtime import LAYOUT_BATCH_MINOR, HipSolver

var = sys.argv[1] if len(sys.argv) > 1 else "w8x"
Ns = [int(x) for x in sys.argv[2:]] or [1, 2, 3]
so = g.sls_spec(var).build(**g.VARIANT_KW[var])
B = 65536
gen = torch.Generator(device="cuda").manual_seed(1234)
A = torch.rand((800, B), generator=gen, device="cuda", dtype=torch.float64)
b = torch.rand((100, B), generator=gen, device="cuda", dtype=torch.float64)
x0 = .02 * torch.rand((8, B), generator=gen, device="cuda", dtype=torch.float64)
for N in Ns:
    objs = [HipSolver(so, B=B) for _ in range(N)]
    streams = [torch.cuda.Stream() for _ in range(N)]
    for o in objs:
        for nm, t in (("A", A), ("b", b), ("x", x0)):
            o.set_device(nm, t, LAYOUT_BATCH_MINOR)
        o.synchronize()
    K = 24
    for k in range(2 * N):
        objs[k % N].solve_async(SOLVE_ARGS["mu0"], SOLVE_ARGS["maxIter"], SOLVE_ARGS["addEye2Hessian"], stream=streams[k % N].cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(K):
        objs[k % N].solve_async(SOLVE_ARGS["mu0"], SOLVE_ARGS["maxIter"], SOLVE_ARGS["addEye2Hessian"], stream=streams[k % N].cuda_stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%s: %d batches in flight: %.4f ms per batch, %.3g solves/s" % (var, N, dt / K * 1e3, B * K / dt), flush=True)
    for o in objs:
        o.close()
