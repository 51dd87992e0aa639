"""Diagnostic (GPU box): what a Newton iteration of mpc_quadcopter costs when the machine is FULL -- batches of 256 .. 4096
instances stopped by maxIter = 10 / 20 / 40, so that the batch time is throughput and not one straggler; printed as
ms x 256 / (sum of iterations), the figure bench.py's throughput floor is built on (there: one solve of 256 instances).
   python scripts/quad_throughput_probe.py [variant]"""
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np

import __graft_entry__ as g
from tenscalc_amd.problems.mpc_quadcopter import SOLVE_ARGS, mpc_quadcopter, quadcopter_data
from tenscalc_amd.runtime import HipSolver

variant = sys.argv[1] if len(sys.argv) > 1 else "ldsc4w"
spec = mpc_quadcopter(folder=g.GENERATED)
so = spec.build(**g.VARIANT_KW[variant])
names = [n for n, _ in spec.parameters + spec.variables]
NB = 4096
ds = [quadcopter_data(k) for k in range(NB)]
st_ = {n: np.stack([np.asarray(d[n], dtype=float).reshape(-1, order="F") for d in ds]) for n in names}
for B in (256, 512, 1024, 4096):
    obj = HipSolver(so, B=B)
    for n in names:
        obj.set(n, st_[n][:B])
    for mi in (10, 20, 40):
        best = None
        for rep in range(3):
            st, it, tm = obj.solve(**dict(SOLVE_ARGS, maxIter=mi))
            best = tm if best is None else min(best, tm)
        tot = int(np.minimum(it, mi + 1).sum())
        print("B=%5d maxIter=%3d: %8.2f ms, %7d iterations -> %.4f ms per iteration and CU slot (256 slots); one block alone would take %.2f ms"
              % (B, mi, best * 1e3, tot, best * 1e3 * min(B, 256) / tot, (mi + 1) * 0.264), flush=True)
    obj.close()
