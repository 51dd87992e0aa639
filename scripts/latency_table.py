"""GPU box: time of ONE batched solve at small batch sizes -- the latency side of the configs (a receding-horizon
controller solves one instance after the other).  python scripts/latency_table.py"""
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np

import __graft_entry__ as g
from tenscalc_amd.runtime import HipSolver

rows = []
for prob, variants in (("dcmotor", ["strict", "interp8", "ldsm8", "ldsc"]), ("l1l2", ["interp8", "ldsm8", "ldsc4"]), ("quadcopter", ["interp8", "ldsm8", "ldsc4"])):
    if prob == "dcmotor":
        from tenscalc_amd.problems.mpc_dcmotor import SOLVE_ARGS, dcmotor_data as data_fn, mpc_dcmotor as mk
    elif prob == "l1l2":
        from tenscalc_amd.problems.l1l2 import SOLVE_ARGS, l1l2_data as data_fn, l1l2 as mk
    else:
        from tenscalc_amd.problems.mpc_quadcopter import SOLVE_ARGS, quadcopter_data as data_fn, mpc_quadcopter as mk
    spec = mk(folder=g.GENERATED)
    names = [n for n, _ in spec.parameters + spec.variables]
    for B in (1, 64, 256):
        data = [data_fn(100 + k) for k in range(B)]
        stacked = {n: np.stack([np.asarray(d[n], dtype=float).reshape(-1, order="F") for d in data]) for n in names}
        for v in variants:
            so = spec.build(**g.VARIANT_KW[v])
            obj = HipSolver(so, B=B)
            for n in names:
                obj.set(n, stacked[n])
            ts = []
            for _ in range(3):
                st, it, tm = obj.solve(**SOLVE_ARGS)
                ts.append(tm)
            obj.close()
            rows.append((prob, B, v, min(ts) * 1e3, int(it.max()), int((st == 0).sum())))
            print("%-10s B=%4d %-8s %9.2f ms  (max %d iterations, %d solved)" % rows[-1], flush=True)
