"""Diagnostic (GPU box): per-dependency-group wall cycles of a body='interp' build.
python scripts/interp_cycles.py l1l2|dcmotor|quad [B]"""
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np

import __graft_entry__ as g
from tenscalc_amd.runtime import HipSolver

which = sys.argv[1] if len(sys.argv) > 1 else "l1l2"
if which == "l1l2":
    from tenscalc_amd.problems.l1l2 import SOLVE_ARGS, l1l2 as mk, l1l2_data as data
    B = 16384
elif which == "quad":
    from tenscalc_amd.problems.mpc_quadcopter import SOLVE_ARGS, mpc_quadcopter as mk, quadcopter_data as data
    B = 512
else:
    from tenscalc_amd.problems.mpc_dcmotor import SOLVE_ARGS, mpc_dcmotor as mk, dcmotor_data as data
    B = 32768
if len(sys.argv) > 2:
    B = int(sys.argv[2])
spec = mk(folder=g.GENERATED)
kw = dict(g.VARIANT_KW["interp"])
kw["tag"] = "interp_stats"
so = spec.build(stats=True, **kw)
names = [n for n, _ in spec.parameters + spec.variables]
ds = [data(1000 + k) for k in range(min(B, 128))]
obj = HipSolver(so, B=B)
for n in names:
    obj.set(n, np.stack([np.asarray(ds[k % len(ds)][n], dtype=float).reshape(-1, order="F") for k in range(B)]))
for _ in range(2):
    st, it, tm = obj.solve(**SOLVE_ARGS)
s = obj.stats()
G = (s.shape[0] - 2) // 2
tot = s[2 * G].mean()
print(f"{which}: B={B} kernel {tm*1e3:.2f} ms, mean cycles/thread {tot:.0f}, iters {it.mean():.2f}, minF trials {s[2*G+1].mean():.1f}")
T = spec.tape
rows = [(s[k].mean(), s[G + k].mean(), k + 1, len(T.groups[k])) for k in range(G)]
rows.sort(reverse=True)
acc = 0
print(" grp  instrs  calls   cycles  cyc/call cyc/instr share")
for cyc, cnt, gid, ni in rows[:30]:
    acc += cyc
    print(f"{gid:4d} {ni:7d} {cnt:6.1f} {cyc:9.0f} {cyc/max(cnt,1e-9):9.0f} {cyc/max(cnt,1e-9)/ni:8.1f} {100*cyc/tot:6.1f}%")
print(f"top groups {100*acc/tot:.1f}% of kernel cycles")
