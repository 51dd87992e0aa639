"""GPU box: the ldsm back-end (one instance per workgroup, hot state in LDS) against the CPU oracle.
   python scripts/check_ldsm.py <sls|dcmotor|l1l2|quad> <variant: ldsm|ldsm4|interp...> [B]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.getcwd())
import __graft_entry__ as g  # noqa: E402
from tenscalc_amd.runtime import HipSolver  # noqa: E402
from tenscalc_amd.tapeio import save_tape  # noqa: E402
from oracle.tc_oracle import Oracle  # noqa: E402
from oracle import build as ob  # noqa: E402

prob, variant = sys.argv[1], sys.argv[2]
B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
g.build_native_table()
ob.build_oracle()
if prob == "sls":
    from tenscalc_amd.problems.sls import SOLVE_ARGS, sls_data
    spec = g.sls_spec("strict")
    data = []
    for k in range(B):
        A, b, x0 = sls_data(3000 + k)
        data.append(dict(A=A, b=b, x=x0))
elif prob == "dcmotor":
    from tenscalc_amd.problems.mpc_dcmotor import SOLVE_ARGS, dcmotor_data, mpc_dcmotor
    spec = mpc_dcmotor(folder=g.GENERATED)
    data = [dcmotor_data(4000 + k) for k in range(B)]
elif prob == "l1l2":
    from tenscalc_amd.problems.l1l2 import SOLVE_ARGS, l1l2, l1l2_data
    spec = l1l2(folder=g.GENERATED)
    data = [l1l2_data(10 + k) for k in range(B)]
else:
    from tenscalc_amd.problems.mpc_quadcopter import SOLVE_ARGS, mpc_quadcopter, quadcopter_data
    spec = mpc_quadcopter(folder=g.GENERATED)
    data = [quadcopter_data(20 + k) for k in range(B)]
t0 = time.time()
so = spec.build(**g.VARIANT_KW[variant])
print("built", so, "%.1fs" % (time.time() - t0), flush=True)
names = [n for n, _ in spec.parameters + spec.variables]
stacked = {n: np.stack([np.asarray(d[n], dtype=float).reshape(-1, order="F") for d in data]) for n in names}
obj = HipSolver(so, B=B, device=0)
for n in names:
    obj.set(n, stacked[n])
st, it, tm = obj.solve(**SOLVE_ARGS)
st, it, tm = obj.solve(**SOLVE_ARGS)
got = np.concatenate([a.reshape(B, -1) for a in obj.getOutputs()], axis=1)
print(prob, variant, "B", B, "kernel %.3f ms" % (tm * 1e3), "status0", int((st == 0).sum()), "iters", it.min(), np.median(it), it.max(), flush=True)
if os.environ.get("TC_NO_ORACLE"):
    sys.exit(0)
tape = "/tmp/%s.tape" % prob
save_tape(spec.tape, tape)
c = spec.classname
nref = min(B, int(os.environ.get("TC_NREF", "64")))
t0 = time.time()
ref, rst, rit = Oracle(tape).solve_batch({f"{c}_set_{n}": stacked[n][:nref] for n in names}, c + "_getOutputs", nref,
                                         mu0=SOLVE_ARGS["mu0"], maxIter=SOLVE_ARGS["maxIter"])
same = (st[:nref] == rst) & (it[:nref] == rit) & np.array([np.array_equal(got[k], ref[k], equal_nan=True) for k in range(nref)])
print("oracle %.1fs; identical %d / %d; status/iter equal %d; max|diff| %.3g" %
      (time.time() - t0, same.sum(), nref, int(((st[:nref] == rst) & (it[:nref] == rit)).sum()), np.nanmax(np.abs(got[:nref] - ref))))
if not same.all():
    k = int(np.where(~same)[0][0])
    print(" first mismatch: instance", k, "status", st[k], rst[k], "iter", it[k], rit[k], "nbad outputs", int((got[k] != ref[k]).sum()))
sys.exit(0 if same.all() else 1)
