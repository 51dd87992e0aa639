"""GPU box: bitwise comparison of a lowering variant of the sls class with the CPU oracle (status, iteration
count, outputs) on seeded instances incl. NaN / infeasible / maxIter-limited ones.
   python scripts/check_exact.py [variant] [B]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402
from tenscalc_amd.problems.sls import SOLVE_ARGS, sls_batch  # noqa: E402
from tenscalc_amd.runtime import HipSolver  # noqa: E402
from tenscalc_amd.tapeio import save_tape  # noqa: E402
from oracle.tc_oracle import Oracle  # noqa: E402
from oracle import build as ob  # noqa: E402

variant = sys.argv[1] if len(sys.argv) > 1 else "w8x"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
g.build_native_table()
ob.build_oracle()
spec = g.sls_spec(variant)
so = spec.build(**g.VARIANT_KW[variant])
ref_spec = g.sls_spec("strict")
tape = "/tmp/sls_ref.tape"
save_tape(ref_spec.tape, tape)
A, b, x0 = sls_batch(B, seed0=20000)
A[3] *= np.nan
b[5] *= 1e6
A[7] *= 0
x0[9] = .06
x0[11] = .05 - 1e-13
c = ref_spec.classname
orc = Oracle(tape)
bad = 0
for kw in [dict(), dict(maxIter=4), dict(mu0=1e-3), dict(addEye2Hessian=(-1.0, -1.0))]:
    args = {**SOLVE_ARGS, **kw}
    obj = HipSolver(so, B=B, device=0)
    obj.setP_A(np.ascontiguousarray(np.transpose(A, (0, 2, 1))))
    obj.setP_b(b)
    obj.setV_x(x0)
    st, it, t = obj.solve(**args)
    J, x = obj.getOutputs()
    t0 = time.time()
    ref, rst, rit = orc.solve_batch({c + "_set_A": np.transpose(A, (0, 2, 1)).reshape(B, -1), c + "_set_b": b,
                                     c + "_set_x": x0}, c + "_getOutputs", B, mu0=args["mu0"], maxIter=args["maxIter"],
                                    addEye2Hessian=args["addEye2Hessian"])
    got = np.concatenate([J.reshape(B, 1), x.reshape(B, 8)], axis=1)
    same = (st == rst) & (it == rit) & np.array([np.array_equal(got[k], ref[k], equal_nan=True) for k in range(B)])
    print(variant, kw, "identical: %d / %d" % (same.sum(), B), "status!=0:", int((rst != 0).sum()),
          "max|dx| %.3g" % np.nanmax(np.abs(got - ref)), "oracle %.1fs" % (time.time() - t0), flush=True)
    if not same.all():
        k = int(np.where(~same)[0][0])
        print("  first mismatch: instance", k, "status", st[k], rst[k], "iter", it[k], rit[k], got[k] - ref[k])
        bad += 1
sys.exit(1 if bad else 0)
