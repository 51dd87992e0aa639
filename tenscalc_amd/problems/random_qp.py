"""Random sparse convex QPs  min 1/2 x'Px + q'x  s.t.  Ex + e = 0,  Cx + d >= 0  with constant sparse P, E, C
(structure and values drawn from a seed) and q, e, d as parameters.  Not a reference example: test problems whose
tapes have IRREGULAR structure -- families of odd lengths, reductions with ragged tails, KKT patterns no hand has
arranged -- for the lowerings to be checked against the oracle on (tests/test_gpu_random_qp.py)."""
from __future__ import annotations

import numpy as np

from .. import tcalculus as tc
from ..cmex2optimizeCS import cmex2optimizeCS

SOLVE_ARGS = dict(mu0=1.0, maxIter=60, saveIter=-1, addEye2Hessian=(1e-9, 1e-9))
SHAPES = {1: (14, 4, 9), 2: (23, 7, 17), 3: (9, 0, 13)}       # seed -> (n, nG, nF)


def random_qp_matrices(seed):
    n, nG, nF = SHAPES[seed]
    rs = np.random.RandomState(1000 + seed)
    M = np.zeros((n, n))
    for k in range(n - 1):                      # banded + a few random couplings: sparse, positive definite
        M[k, k + 1] = M[k + 1, k] = .3 * rs.randn()
    for _ in range(n // 3):
        i, j = rs.randint(n, size=2)
        if i != j:
            M[i, j] = M[j, i] = .2 * rs.randn()
    P = M + np.diag(np.abs(M).sum(1) + .5 + rs.rand(n))
    E = np.zeros((nG, n))
    for r in range(nG):
        E[r, rs.choice(n, 3, replace=False)] = rs.randn(3)
    C = np.zeros((nF, n))
    for r in range(nF):
        C[r, rs.choice(n, 2, replace=False)] = rs.randn(2)
    return P, E, C


def random_qp(seed, classname=None, **options):
    P, E, C = random_qp_matrices(seed)
    n, nG, nF = SHAPES[seed]
    x = tc.Tvariable("x", (n,))
    q = tc.Tvariable("q", (n,))
    d = tc.Tvariable("d", (nF,))
    J = .5 * tc.tprod(x, (-1,), tc.Tconstant(P) * x, (-1,)) + tc.tprod(q, (-1,), x, (-1,))
    cons, pars = [], [q]
    if nG:
        e = tc.Tvariable("e", (nG,))
        cons.append(tc.Tconstant(E) * x + e == 0)
        pars.append(e)
    cons.append(tc.Tconstant(C) * x + d >= 0)
    pars.append(d)
    return cmex2optimizeCS(classname=classname or "tmp_rqp%d" % seed, objective=J, optimizationVariables=[x],
                           constraints=cons, parameters=pars, outputExpressions=[J, x], outputNames=["J", "x"], **options)


def random_qp_data(seed, B, seed0=0):
    """B instances: q random, a strictly feasible start x0, e / d placed so that x0 nearly satisfies the equalities and
    strictly satisfies the inequalities"""
    P, E, C = random_qp_matrices(seed)
    n, nG, nF = SHAPES[seed]
    rs = np.random.RandomState(seed0 + 77 * seed)
    x0 = .5 * rs.randn(B, n)
    out = dict(q=rs.randn(B, n), x=x0, d=-x0 @ C.T + .5 + rs.rand(B, nF))
    if nG:
        out["e"] = -x0 @ E.T + .05 * rs.randn(B, nG)
    return out
