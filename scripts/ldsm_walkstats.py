"""GPU box (diagnostic): split of the ldsm record walk's cycles for wave 0 of instance 0 -- waiting for body words,
executing records, barriers, the rest.  python scripts/ldsm_walkstats.py [quad|dcmotor] [ldsm4|ldsm] [B]"""
import ctypes
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np

import __graft_entry__ as g
from tenscalc_amd.runtime import HipSolver

prob = sys.argv[1] if len(sys.argv) > 1 else "quad"
variant = sys.argv[2] if len(sys.argv) > 2 else "ldsm4"
B = int(sys.argv[3]) if len(sys.argv) > 3 else 4
if prob == "dcmotor":
    from tenscalc_amd.problems.mpc_dcmotor import SOLVE_ARGS, dcmotor_data as data_fn, mpc_dcmotor as mk
else:
    from tenscalc_amd.problems.mpc_quadcopter import SOLVE_ARGS, quadcopter_data as data_fn, mpc_quadcopter as mk
spec = mk(folder=g.GENERATED)
kw = dict(g.VARIANT_KW[variant])
kw["tag"] = variant + "_ws"
kw["extra_flags"] = ("-DTCL_WALKSTATS=1",)
so = spec.build(**kw)
data = [data_fn(20 + k) for k in range(B)]
names = [n for n, _ in spec.parameters + spec.variables]
obj = HipSolver(so, B=B)
for n in names:
    obj.set(n, np.stack([np.asarray(d[n], dtype=float).reshape(-1, order="F") for d in data]))
st, it, tm = obj.solve(**SOLVE_ARGS)
lib = ctypes.CDLL(so)
out = (ctypes.c_ulonglong * 56)()
rc = lib.tc_ldsm_walkstats(out)
w = [int(x) for x in out]
tot = sum(w[:4])
print(f"{prob} {variant}: kernel {tm*1e3:.1f} ms, instance 0: {it[0]} iterations, status {st[0]}; rc={rc}")
print(f"walk cycles (wave 0): total {tot}  = {tot/max(it[0],1):.0f} per iteration")
for nm, v in zip(["wait for body words", "execute records", "barriers", "rest of the walk"], w[:4]):
    print(f"  {nm:22s} {v:12d}  {100*v/tot:5.1f}%")
print(f"  of the rest: issuing the body-word loads of the record 4 ahead {w[7]} = {w[7]/max(w[4]+w[5],1):.0f} per step")
print(f"  per call: entry barrier {w[48]/max(w[6],1):.0f}, record range + first headers {w[49]/max(w[6],1):.0f}")
print(f"records {w[4]} ({w[1]/max(w[4],1):.0f} cycles each), barriers {w[5]} ({w[2]/max(w[5],1):.0f} each), "
      f"calls {w[6]} ({w[3]/max(w[6],1):.0f} rest each); wait per step {w[0]/max(w[4]+w[5],1):.0f}")
names_ = ["long"] + ["short n=%d" % n for n in range(1, 9)] + ["fold <=%d" % (8 * (q + 1)) for q in range(8)] + ["fold >64"]
print("by record class (cycles include the atomics of this diagnostic):")
for q, nm in enumerate(names_):
    if w[28 + q]:
        print(f"  {nm:12s} {w[28+q]:8d} records  {w[8+q]/w[28+q]:8.0f} cycles each  {100*w[8+q]/tot:5.1f}% of the walk")
