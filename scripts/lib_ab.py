"""GPU box: one config's batch through several PREBUILT class libraries (by tag), timed and compared bit for bit with the
first.  python scripts/lib_ab.py dcmotor|l1l2|quadcopter B tag [tag ...]"""
import importlib
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np

import __graft_entry__ as g
from tenscalc_amd.runtime import HipSolver

MODS = {"dcmotor": ("tenscalc_amd.problems.mpc_dcmotor", "mpc_dcmotor", "dcmotor_data"),
        "l1l2": ("tenscalc_amd.problems.l1l2", "l1l2", "l1l2_data"),
        "quadcopter": ("tenscalc_amd.problems.mpc_quadcopter", "mpc_quadcopter", "quadcopter_data")}
which, B, tags = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
if which == "sls":
    mod = importlib.import_module("tenscalc_amd.problems.sls")
    spec = g.sls_spec("w8x")
    A_, b_, x_ = mod.sls_batch(min(B, 4096))
    rep = -(-B // A_.shape[0])
    host = {"A": np.tile(np.ascontiguousarray(np.transpose(A_, (0, 2, 1))).reshape(A_.shape[0], -1), (rep, 1))[:B],
            "b": np.tile(b_, (rep, 1))[:B], "x": np.tile(x_, (rep, 1))[:B]}
else:
    mod = importlib.import_module(MODS[which][0])
    spec = getattr(mod, MODS[which][1])(folder=g.GENERATED)
    data = getattr(mod, MODS[which][2])
    ds = [data(k) for k in range(B)]
    host = {n: np.stack([np.asarray(x[n], dtype=float).reshape(-1, order="F") for x in ds])
            for n in [n for n, _ in spec.parameters + spec.variables]}
d = os.path.join(g.GENERATED, "@" + spec.classname)
names = [n for n, _ in spec.parameters + spec.variables]
ref = None
for tag in tags:
    obj = HipSolver(os.path.join(d, "lib%s_%s.so" % (spec.classname, tag)), B=B)
    for n in names:
        obj.set(n, host[n])
    ts = []
    for _ in range(9):
        st, it, tm = obj.solve(**mod.SOLVE_ARGS)
        ts.append(tm)
    out = np.concatenate([o.reshape(B, -1) for o in obj.getOutputs()], axis=1)
    obj.close()
    if ref is None:
        ref = (st, it, out)
    same = np.array_equal(st, ref[0]) and np.array_equal(it, ref[1]) and np.array_equal(out, ref[2], equal_nan=True)
    print("%s %s B=%d: min %.3f ms, median %.3f ms; status0 %.4f, iterations median %d max %d; bitwise equal to the first: %s"
          % (which, tag, B, min(ts) * 1e3, np.median(ts) * 1e3, np.mean(st == 0), np.median(it), it.max(), same), flush=True)
