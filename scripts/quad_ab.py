"""GPU box: mpc_quadcopter B=512 through two prebuilt ldsm8 libraries (fused factor+solve walk vs not), timed and
compared bit for bit.  python scripts/quad_ab.py"""
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np

import __graft_entry__ as g
from tenscalc_amd.problems.mpc_quadcopter import SOLVE_ARGS, mpc_quadcopter, quadcopter_data
from tenscalc_amd.runtime import HipSolver

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
spec = mpc_quadcopter(folder=g.GENERATED)
d = os.path.join(g.GENERATED, "@" + spec.classname)
names = [n for n, _ in spec.parameters + spec.variables]
ds = [quadcopter_data(k) for k in range(B)]
host = {n: np.stack([np.asarray(x[n], dtype=float).reshape(-1, order="F") for x in ds]) for n in names}
ref = None
for tag in ("ldsm8_nofuse", "ldsm8"):
    so = os.path.join(d, "lib%s_%s.so" % (spec.classname, tag))
    obj = HipSolver(so, B=B)
    for n in names:
        obj.set(n, host[n])
    ts = []
    for _ in range(3):
        st, it, tm = obj.solve(**SOLVE_ARGS)
        ts.append(tm)
    out = np.concatenate([o.reshape(B, -1) for o in obj.getOutputs()], axis=1)
    obj.close()
    if ref is None:
        ref = (st, it, out)
    same = np.array_equal(st, ref[0]) and np.array_equal(it, ref[1]) and np.array_equal(out, ref[2], equal_nan=True)
    print("%s B=%d: min %.1f ms, median %.1f ms; iterations median %d max %d; %.3f ms per iteration of the slowest; "
          "bitwise equal to the first: %s" % (tag, B, min(ts) * 1e3, np.median(ts) * 1e3, np.median(it), it.max(),
                                              min(ts) * 1e3 / it.max(), same), flush=True)
