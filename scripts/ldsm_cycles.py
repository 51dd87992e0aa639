"""GPU box: where does an ldsm solve spend its cycles?  python scripts/ldsm_cycles.py <quad|dcmotor|l1l2> [variant] [B]"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import __graft_entry__ as g
from tenscalc_amd.runtime import HipSolver
prob = sys.argv[1]; variant = sys.argv[2] if len(sys.argv) > 2 else "ldsm4"; B = int(sys.argv[3]) if len(sys.argv) > 3 else 16
if prob == "dcmotor":
    from tenscalc_amd.problems.mpc_dcmotor import SOLVE_ARGS, dcmotor_data as data_fn, mpc_dcmotor as mk
elif prob == "l1l2":
    from tenscalc_amd.problems.l1l2 import SOLVE_ARGS, l1l2_data as data_fn, l1l2 as mk
else:
    from tenscalc_amd.problems.mpc_quadcopter import SOLVE_ARGS, quadcopter_data as data_fn, mpc_quadcopter as mk
spec = mk(folder=g.GENERATED)
kw = dict(g.VARIANT_KW[variant]); kw["tag"] = variant + "_stats"
so = spec.build(stats=True, **kw)
data = [data_fn(20 + k) for k in range(B)]
names = [n for n, _ in spec.parameters + spec.variables]
obj = HipSolver(so, B=B)
for n in names:
    obj.set(n, np.stack([np.asarray(d[n], dtype=float).reshape(-1, order="F") for d in data]))
st, it, tm = obj.solve(**SOLVE_ARGS)
s = obj.stats()
G = (s.shape[0] - 10) // 2
k = int(np.argmin(np.abs(it - np.median(it))))      # a typical instance
tot = s[2 * G, k]
print(f"{prob} {variant}: kernel {tm*1e3:.1f} ms; instance {k}: {it[k]} iterations, {tot:.0f} cycles total = {tot/it[k]:.0f} per iteration")
print("phases [init, exit tests, curvature+direction error, alpha bounds, line search, update, mu]:", " ".join("%.0f" % (s[2 * G + 2 + q, k] / it[k]) for q in range(7)), "per iteration")
T = spec.tape
from tenscalc_amd.ldsm_schedule import group_levels
lvl, nlev = group_levels(T)
rows = sorted(((s[q, k], s[G + q, k], q + 1) for q in range(G)), reverse=True)[:14]
print(" grp  instrs levels  calls   cycles  cyc/call  cyc/level share")
for cyc, cnt, gid in rows:
    print(f"{gid:4d} {len(T.groups[gid-1]):7d} {nlev[gid]:6d} {cnt:6.0f} {cyc:9.0f} {cyc/max(cnt,1):9.0f} {cyc/max(cnt,1)/max(nlev[gid],1):8.0f} {100*cyc/tot:6.1f}%")
