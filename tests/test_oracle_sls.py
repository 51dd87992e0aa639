"""CPU: the oracle on sls N=100 n=8 against (1) the committed golden vectors, (2) the
tape-independent dense closed form, (3) scipy's bounded least squares (mirrors the quadprog
cross-check of examples/sls.m:172-181)."""
import os

import numpy as np
import pytest

from tenscalc_amd.problems.sls import SOLVE_ARGS, sls_data

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sls_N100_n8.npz")


def _solve(o, c, seed):
    A, b, x0 = sls_data(seed)
    o.set(c + "_set_A", A)
    o.set(c + "_set_b", b)
    o.set(c + "_set_x", x0)
    st, it = o.solve(SOLVE_ARGS["mu0"], SOLVE_ARGS["maxIter"], -1, SOLVE_ARGS["addEye2Hessian"])
    return st, it, o.get(c + "_getOutputs"), (A, b, x0)


def test_tape_shape(sls_spec):
    T = sls_spec.tape
    d = sls_spec.defines
    assert (d["nU"], d["nG"], d["nF"]) == (8, 0, 16)
    st = T.ldl_stats[0]
    assert st["n"] == 24 and st["nnzA"] == 112 and st["nonsym"] == 0
    names = {e["name"] for e in T.sets + T.gets + T.copies}
    for need in ["initPrimal__", "scaleIneq__", "setMu__", "initDualIneq__", "setAddEye2HessianU__",
                 "setAddEye2HessianEq__", "getNorminf_Grad__", "getGapMinFMinLambda__", "getCurvature__",
                 "getDirectionError__", "getMaxAlphas_s__", "getMinF_s__", "setAlphaPrimal__",
                 "setAlphaDualIneq__", "updatePrimalDual__", "getJ__", "getGrad__", "getHessInertia__"]:
        assert need in names, need                              # ipmPD_CSsolver.c:28-51


def test_golden_vectors(sls_spec, sls_tape_path):
    from oracle.tc_oracle import Oracle
    g = np.load(GOLD)
    o = Oracle(sls_tape_path)
    c = sls_spec.classname
    for k, seed in enumerate(g["seeds"]):
        st, it, y, (A, b, x0) = _solve(o, c, int(seed))
        assert A.sum() + 2 * b.sum() + 3 * x0.sum() == g["checksum"][k]
        assert st == g["status"][k] and it == g["iter"][k]
        assert y[0] == g["J"][k] and np.array_equal(y[1:], g["x"][k])


def test_against_dense_closed_form(sls_spec, sls_tape_path):
    """Pins TAPE CONSTRUCTION: iterates of the tape-driven solver and of the dense numpy closed
    form agree to rounding while no line-search decision sits on a knife edge.  (The reference
    algorithm's first trial step alphaMax_/stepback lands *exactly* on the constraint boundary
    whenever the primal step is the limiting one -- ipmPD_CSsolver.c:693-695 after :686 -- so the
    sign of a 1e-17 residue picks the branch; after such a step two correct implementations
    differ by ~1e-5 in the iterate and ~1e-8 in the solution.  Hence: tight tolerance on the
    first iterations, solver tolerance on the final point.)"""
    from oracle.tc_oracle import Oracle
    from sls_closed_form import solve_sls_dense
    o = Oracle(sls_tape_path)
    c = sls_spec.classname
    for seed in range(12):
        A, b, x0 = sls_data(seed)
        for k in range(1, 6):          # state after k full Newton iterations
            o.set(c + "_set_A", A); o.set(c + "_set_b", b); o.set(c + "_set_x", x0)
            st, it = o.solve(1.0, k, -1, (1e-9, 1e-9))
            y = o.get(c + "_getOutputs")
            st2, it2, u2, tr = solve_sls_dense(A, b, x0, maxIter=k)
            assert it == it2 == k + 1 and st2 == 8 and st & 8
            np.testing.assert_allclose(y[1:], u2, rtol=0, atol=1e-13)
            np.testing.assert_allclose(y[0], ((A @ u2 - b) ** 2).sum() / 100, rtol=1e-13)
        st, it, y, _ = _solve(o, c, seed)
        st2, it2, u2, _ = solve_sls_dense(A, b, x0)
        assert st == st2 == 0 and abs(it - it2) <= 1
        np.testing.assert_allclose(y[1:], u2, rtol=0, atol=2e-6)


def test_against_scipy_bounded_least_squares(sls_spec, sls_tape_path):
    from scipy.optimize import lsq_linear
    from oracle.tc_oracle import Oracle
    o = Oracle(sls_tape_path)
    c = sls_spec.classname
    for seed in range(8):
        st, it, y, (A, b, x0) = _solve(o, c, seed)
        r = lsq_linear(A, b, bounds=(0, .05), tol=1e-12)
        assert st == 0
        assert np.abs(y[1:] - r.x).max() < 1e-4          # limited by gradTolerance / desiredDualityGap
        assert y[0] - ((A @ r.x - b) ** 2).sum() / 100 < 1e-5


def test_flop_accounting_and_lazy_groups(sls_spec, sls_tape_path):
    """parameter-only groups (A'A) are evaluated once per set, not once per iteration
    (doc/csparse.tex:176-184)."""
    from oracle.tc_oracle import Oracle
    o = Oracle(sls_tape_path)
    c = sls_spec.classname
    st, it, y, _ = _solve(o, c, 0)
    cnt = o.counters()
    assert 40_000 < cnt["flops"] < 120_000 and 10 <= cnt["minF_trials"] <= 40
    T = sls_spec.tape
    big = int(np.argmax(T.flops.sum(axis=1)))      # the A'A group
    per_iter = cnt["flops"] - int(T.flops[big].sum())
    assert per_iter < it * 8000
    o.reset_counters()
    st2, it2 = o.solve(SOLVE_ARGS["mu0"], SOLVE_ARGS["maxIter"], -1, SOLVE_ARGS["addEye2Hessian"])
    assert (st2, it2) == (st, it)
    f2 = o.counters()["flops"]                    # second solve: A'A (and other parameter-only groups) still valid
    assert per_iter - 2000 < f2 <= per_iter


def test_status_bits_maxiter_and_infeasible_start(sls_spec, sls_tape_path):
    from oracle.tc_oracle import Oracle
    o = Oracle(sls_tape_path)
    c = sls_spec.classname
    A, b, x0 = sls_data(0)
    o.set(c + "_set_A", A); o.set(c + "_set_b", b); o.set(c + "_set_x", x0)
    st, it = o.solve(1.0, 3, -1, (1e-9, 1e-9))
    assert it == 4 and st & 8 and st & 64 and st & 128     # maxIter reached, large gap, large mu
    o.set(c + "_set_x", -x0)                                # violates x>=0 -> scaling makes F=+1... still interior?
    st, it = o.solve(1.0, 30, -1, (1e-9, 1e-9))
    assert st in (0, 1, 2, 4) or st & 8
    o.set(c + "_set_A", A * np.nan)
    st, it = o.solve(1.0, 30, -1, (1e-9, 1e-9))
    assert st == 4 and it == 1                              # NaN -> "failed to invert hessian"
