"""Tape construction on RANDOM convex QPs: sparse cost, sparse equality and inequality constraints with random patterns,
written with tcalculus the way a user would (x'Px, C*x+d >= 0, E*x+e == 0 with sparse CONSTANT matrices), against the
dense numpy interior-point method of tests/dense_ipm.py working on the same numbers.  Same status and iteration count,
iterates to 1e-7 after 1-3 Newton steps -- for the greedy elimination order and for the multiple-elimination orders
(tolerance 0 / 1), whose numerics on KKT matrices with equality duals are the delicate case."""
import numpy as np
import pytest

from dense_ipm import solve_qp_dense


def _random_qp(seed, n=14, nG=4, nF=9):
    rs = np.random.RandomState(seed)
    M = np.zeros((n, n))
    for k in range(n - 1):                      # banded + a few random couplings: sparse, positive definite
        M[k, k + 1] = M[k + 1, k] = .3 * rs.randn()
    for _ in range(4):
        i, j = rs.randint(n, size=2)
        if i != j:
            M[i, j] = M[j, i] = .2 * rs.randn()
    P = M + np.diag(np.abs(M).sum(1) + .5 + rs.rand(n))
    q = rs.randn(n)
    E = np.zeros((nG, n))
    for r in range(nG):
        E[r, rs.choice(n, 3, replace=False)] = rs.randn(3)
    x0 = .5 * rs.randn(n)
    e = -E @ x0 + .05 * rs.randn(nG)
    C = np.zeros((nF, n))
    for r in range(nF):
        C[r, rs.choice(n, 2, replace=False)] = rs.randn(2)
    d = -C @ x0 + .5 + rs.rand(nF)              # x0 strictly feasible for the inequalities
    return P, q, E, e, C, d, x0


@pytest.mark.parametrize("order", ["mindeg", ("mmd", 0), ("mmd", 1)])
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_qp_tape_vs_dense(native_built, tape_oracle, tmp_path, seed, order):
    from tenscalc_amd import tcalculus as tc
    from tenscalc_amd.cmex2optimizeCS import cmex2optimizeCS
    P, q, E, e, C, d, x0 = _random_qp(seed)
    n = len(q)
    x = tc.Tvariable("x", (n,))
    qv = tc.Tvariable("q", (n,))                                   # a parameter, so that something is set from outside
    Pc, Ec, Cc = tc.Tconstant(P), tc.Tconstant(E), tc.Tconstant(C)
    J = .5 * tc.tprod(x, (-1,), Pc * x, (-1,)) + tc.tprod(qv, (-1,), x, (-1,))
    spec = cmex2optimizeCS(classname="tmp_rqp_%d" % seed, objective=J, optimizationVariables=[x],
                           constraints=[Ec * x + tc.Tconstant(e) == 0, Cc * x + tc.Tconstant(d) >= 0],
                           parameters=[qv], outputExpressions=[J, x], outputNames=["J", "x"], folder=str(tmp_path),
                           ldlOrder=order)
    assert (spec.defines["nU"], spec.defines["nG"], spec.defines["nF"]) == (n, E.shape[0], C.shape[0])
    c = spec.classname
    o = tape_oracle(spec.tape, c)
    for mi in (1, 2, 3, 60):
        o.set(c + "_set_q", q)
        o.set(c + "_set_x", x0)
        st, it = o.solve(1.0, mi, -1, (1e-9, 1e-9))
        y = o.get(c + "_getOutputs")
        dst, dit, u, _, _ = solve_qp_dense(P, q, 0.0, E, e, C, d, x0, 1.0, mi)
        assert (st & 15, it) == (dst, dit), (mi, st, it, dst, dit)
        tol = 1e-7 if mi <= 3 else 1e-5
        assert np.abs(y[1:] - u).max() <= tol * (1 + np.abs(u).max()), (mi, np.abs(y[1:] - u).max())
        assert abs(y[0] - (.5 * u @ P @ u + q @ u)) <= 1e-6 * (1 + abs(y[0]))
    assert st == 0
