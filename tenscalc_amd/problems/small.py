"""Small least-squares problems that exercise the branches of the solver template the BASELINE
problems do not: equality + inequality constraints with inertia-based regularisation and
uncoupled step lengths, equality constraints only (nF==0), and no constraints at all."""
from __future__ import annotations

from .. import tcalculus as tc
from ..cmex2optimizeCS import cmex2optimizeCS

KINDS = {
    "eq": dict(useInertia=True, coupledAlphas=False, scaleCost=10.0),
    "eqonly": dict(),
    "free": dict(adjustAddEye2Hessian=False),
}


def small_ls(kind, N=12, n=4, **options):
    A = tc.Tvariable("A", (N, n))
    b = tc.Tvariable("b", (N,))
    x = tc.Tvariable("x", (n,))
    J = tc.norm2(A * x - b) / N
    cons = {"eq": [tc.tsum(x) == .1, x >= 0], "eqonly": [tc.tsum(x) == .1], "free": []}[kind]
    opt = dict(KINDS[kind])
    opt.update(options)
    return cmex2optimizeCS(classname="tmp_small_" + kind, objective=J, optimizationVariables=[x],
                           constraints=cons, outputExpressions=[J, x], outputNames=["J", "x"],
                           parameters=[A, b], **opt)
