"""Where does a batch solve spend its time?  Kernel time of an sls class for maxIter = 1, 2, 4, 8, 30:
the intercept is prologue + epilogue (inputs streamed, A'A, outputs), the slope the Newton iterations.
   python scripts/iter_time.py [variant]      (GPU box)"""
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np
import torch

import __graft_entry__ as g
from tenscalc_amd.problems.sls import SOLVE_ARGS
from tenscalc_amd.runtime import LAYOUT_BATCH_MINOR, HipSolver

var = sys.argv[1] if len(sys.argv) > 1 else "w8s"
spec = g.sls_spec(var)
so = spec.build(**g.VARIANT_KW[var])
B = 65536
gen = torch.Generator(device="cuda").manual_seed(1234)
A = torch.rand((800, B), generator=gen, device="cuda", dtype=torch.float64)
b = torch.rand((100, B), generator=gen, device="cuda", dtype=torch.float64)
x0 = .02 * torch.rand((8, B), generator=gen, device="cuda", dtype=torch.float64)
obj = HipSolver(so, B=B)
for nm, t in (("A", A), ("b", b), ("x", x0)):
    obj.set_device(nm, t, LAYOUT_BATCH_MINOR)
obj.synchronize()
for mi in (1, 2, 4, 8, 30):
    ts = []
    for _ in range(5):
        st, it, tm = obj.solve(mu0=SOLVE_ARGS["mu0"], maxIter=mi)
        ts.append(tm)
    print(f"{var} maxIter={mi}: {np.median(ts)*1e3:.3f} ms, mean iterations run {np.minimum(it, mi).mean():.2f}", flush=True)
