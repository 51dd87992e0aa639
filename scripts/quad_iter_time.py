"""GPU box: time per Newton iteration of the mpc_quadcopter (T=100) interp build: maxIter capped."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import __graft_entry__ as g
from tenscalc_amd.runtime import HipSolver
from tenscalc_amd.problems.mpc_quadcopter import SOLVE_ARGS, mpc_quadcopter, quadcopter_data
spec = mpc_quadcopter(folder=g.GENERATED)
so = spec.build(**g.VARIANT_KW[os.environ.get("TC_VARIANT", "interp")])
names = [n for n, _ in spec.parameters + spec.variables]
for B in (int(x) for x in (sys.argv[1:] or ["64", "512"])):
    ds = [quadcopter_data(1000 + k) for k in range(min(B, 64))]
    obj = HipSolver(so, B=B)
    for n in names:
        obj.set(n, np.stack([np.asarray(ds[k % len(ds)][n], dtype=float).reshape(-1, order="F") for k in range(B)]))
    args = dict(SOLVE_ARGS); args["maxIter"] = 6
    st, it, t = obj.solve(**args)
    print(f"quad B={B}: {t*1e3:.1f} ms for {it.max()} iterations -> {t*1e3/it.max():.1f} ms/iteration", flush=True)
    del obj
