"""GPU box: compare every tape `get` right after the solver's initialisation sequence between a
class library (tc_probe) and the oracle, for a few instances.  python scripts/probe_diff.py quad|l1l2|dcmotor [variant]"""
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np

import __graft_entry__ as g
from tenscalc_amd.runtime import HipSolver
from tenscalc_amd.tapeio import save_tape
from oracle.tc_oracle import Oracle

which = sys.argv[1] if len(sys.argv) > 1 else "quad"
variant = sys.argv[2] if len(sys.argv) > 2 else "interp"
if which == "l1l2":
    from tenscalc_amd.problems.l1l2 import SOLVE_ARGS, l1l2 as mk, l1l2_data as data
elif which == "quad":
    from tenscalc_amd.problems.mpc_quadcopter import SOLVE_ARGS, mpc_quadcopter as mk, quadcopter_data as data
else:
    from tenscalc_amd.problems.mpc_dcmotor import SOLVE_ARGS, mpc_dcmotor as mk, dcmotor_data as data
B = 2
spec = mk(folder=g.GENERATED)
so = spec.build(**g.VARIANT_KW[variant])
tape = f"/tmp/{which}.tape"
save_tape(spec.tape, tape)
c = spec.classname
names = [n for n, _ in spec.parameters + spec.variables]
ds = [data(1000 + k) for k in range(B)]
obj = HipSolver(so, B=B)
for n in names:
    obj.set(n, np.stack([np.asarray(d[n], dtype=float).reshape(-1, order="F") for d in ds]))
T = spec.tape
mu0 = SOLVE_ARGS["mu0"]
ae = SOLVE_ARGS.get("addEye2Hessian", (1e-9, 1e-9))
D = spec.defines
ref = {}
for k in range(B):
    o = Oracle(tape)
    for n in names:
        o.set(f"{c}_set_{n}", ds[k][n])
    if D["setAddEye2Hessian"]:
        o.set("setAddEye2HessianU__", [ae[0]]); o.set("setAddEye2HessianEq__", [ae[1]])
    o.copy("initPrimal__")
    if D["scaleCost"]:
        o.copy("scaleCost__")
    if D["nF"] > 0:
        if D["scaleInequalities"]:
            o.copy("scaleIneq__")
        o.set("setMu__", [mu0]); o.copy("initDualIneq__")
    if D["nG"] > 0:
        o.copy("initDualEqX__")
    for e in T.gets:
        ref.setdefault(e["name"], []).append(o.get(e["name"]))
bad = 0
for e in T.gets:
    got = obj.probe(e["name"], mu0=mu0, addEye2Hessian=ae)
    r = np.stack(ref[e["name"]])
    eq = np.array_equal(got, r, equal_nan=True)
    d = np.abs(got - r)
    rel = np.nanmax(d / np.maximum(np.abs(r), 1e-300)) if d.size else 0
    print(f"{e['name']:28s} n={r.shape[1]:6d} bit-equal={eq} max|diff|={np.nanmax(d) if d.size else 0:.3g} maxrel={rel:.3g}")
    if not eq:
        bad += 1
        idx = np.argwhere(got != r)[:5]
        for i, j in idx:
            print(f"     inst {i} elem {j}: got {got[i, j]!r} ref {r[i, j]!r}")
print("mismatching gets:", bad)
