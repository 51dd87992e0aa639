"""A/B of class libraries given by PATH (GPU box): the same batch through each, timed and compared bit for bit with the first.
   python scripts/so_ab.py <dcmotor|l1l2|quadcopter> <B> lib1.so lib2.so ...
(kw_sweep.py builds its variants from the working tree; this takes libraries built anywhere -- e.g. from another commit in a
git worktree -- so that two versions of the generator or of csrc/ meet on ONE box.)"""
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np

import __graft_entry__ as g
from tenscalc_amd.runtime import HipSolver

which, B, libs = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
if which == "l1l2":
    from tenscalc_amd.problems.l1l2 import SOLVE_ARGS, l1l2 as mk, l1l2_data as data
elif which == "quadcopter":
    from tenscalc_amd.problems.mpc_quadcopter import SOLVE_ARGS, mpc_quadcopter as mk, quadcopter_data as data
else:
    from tenscalc_amd.problems.mpc_dcmotor import SOLVE_ARGS, dcmotor_data as data, mpc_dcmotor as mk
spec = mk(folder=g.GENERATED)
names = [n for n, _ in spec.parameters + spec.variables]
nd = min(B, int(os.environ.get("TC_AB_DISTINCT", "256")))
ds = [data(k) for k in range(nd)]
host = {n: np.stack([np.asarray(ds[k % nd][n], dtype=float).reshape(-1, order="F") for k in range(B)]) for n in names}
ref = None
for rep in range(2):
    for so in libs:
        obj = HipSolver(so, B=B)
        for n in names:
            obj.set(n, host[n])
        obj.synchronize()
        ts = []
        for _ in range(int(os.environ.get("TC_SWEEP_REPS", "5"))):
            st, it, tm = obj.solve(**SOLVE_ARGS)
            ts.append(tm)
        out = np.concatenate([o.reshape(B, -1) for o in obj.getOutputs()], axis=1)
        obj.close()
        if ref is None:
            ref = (st, it, out)
        same = np.array_equal(st, ref[0]) and np.array_equal(it, ref[1]) and np.array_equal(out, ref[2], equal_nan=True)
        print("%s B=%d %-60s min %.4f median %.4f ms  bitwise equal to the first: %s" % (
            which, B, os.path.basename(so), min(ts) * 1e3, np.median(ts) * 1e3, same), flush=True)
