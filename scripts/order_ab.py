"""A/B of LDL elimination orders on the GPU box: python scripts/order_ab.py <dcmotor|l1l2|quad> <variant> <B> <order>...
an order is mindeg | mmdK | mmdK_two | mmdK_allbut1 | depthW.  `--build-only` as last argument compiles the libraries and exits (CPU container)."""
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np

import __graft_entry__ as g

args = [a for a in sys.argv[1:] if a != "--build-only"]
build_only = "--build-only" in sys.argv
which, variant, B = args[0], args[1], int(args[2])
if which == "l1l2":
    from tenscalc_amd.problems.l1l2 import SOLVE_ARGS, l1l2 as mk, l1l2_data as data
elif which == "quad":
    from tenscalc_amd.problems.mpc_quadcopter import SOLVE_ARGS, mpc_quadcopter as mk, quadcopter_data as data
else:
    from tenscalc_amd.problems.mpc_dcmotor import SOLVE_ARGS, mpc_dcmotor as mk, dcmotor_data as data


def order(nm):
    if nm == "mindeg":
        return "mindeg"
    if nm.startswith("mmd"):          # mmd<tolerance>[_<gate>]
        t, _, gate = nm[3:].partition("_")
        return ("mmd", int(t), gate) if gate else ("mmd", int(t))
    return ("depth", int(nm[5:]))


for nm in args[3:]:
    spec = mk(folder=g.GENERATED, classname="tmpo_%s_%s" % (which, nm), ldlOrder=order(nm))
    so = spec.build(**g.VARIANT_KW[variant])
    if build_only:
        print("built", so, flush=True)
        continue
    from tenscalc_amd.runtime import HipSolver
    names = [n for n, _ in spec.parameters + spec.variables]
    nd = min(B, 256)
    ds = [data(k) for k in range(nd)]
    obj = HipSolver(so, B=B)
    for n in names:
        obj.set(n, np.stack([np.asarray(ds[k % nd][n], dtype=float).reshape(-1, order="F") for k in range(B)]))
    ts = []
    for _ in range(3):
        st, it, t = obj.solve(**SOLVE_ARGS)
        ts.append(t)
    obj.close()
    print("%s %s B=%d order=%-8s %9.2f ms  %.3g solves/s  iters median %d max %d  solved %d  (ldl %s)" % (
        which, variant, B, nm, min(ts) * 1e3, B / min(ts), np.median(it), it.max(), int((st == 0).sum()),
        [(d["n"], d["nnzL"]) for d in spec.tape.ldl_stats]), flush=True)
