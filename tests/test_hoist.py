"""Opt-in rewrite of tenscalc_amd/hoist.py: parameter-only factors are contracted first.  The
rewritten expressions must evaluate to the same numbers (to rounding) as the originals through
the same scalarizer + CPU oracle, and must lower the per-iteration work."""
import numpy as np
import pytest

import tenscalc_amd.tcalculus as tc
from tenscalc_amd.hoist import Hoister
from test_csparse_ops import run

EPS = np.finfo(float).eps


def test_least_squares_gradient_is_hoisted(tape_oracle):
    rs = np.random.RandomState(3)
    A = tc.Tvariable("A", (11, 4)); b = tc.Tvariable("b", (11,)); x = tc.Tvariable("x", (4,))
    J = tc.norm2(A * x - b) / 11
    H, g = tc.hessian(J, x)
    h = Hoister([A, b])
    g2, H2 = h.rewrite(g), h.rewrite(H)
    assert h.fast_cost(g2) * 3 < h.fast_cost(g)          # 2 passes over A -> one 4x4 product
    assert h.is_slow(H2)                                 # parameter-only, as before (canonical form may differ)
    assert [k for k, *_ in h.log] == ["reassociate", "distribute"] or len(h.log) >= 2
    # the x-dependent part of the rewritten gradient no longer touches A or b directly
    def direct_fast_users(f, seen):
        if f.idx in seen or h.is_slow(f):
            return set()
        seen.add(f.idx)
        u = {a.par for a in f.args if a.op == "variable" and a.idx in h.slow_vars}
        for a in f.args:
            u |= direct_fast_users(a, seen)
        return u
    assert direct_fast_users(g2, set()) == set()
    assert direct_fast_users(g, set()) == {"A", "b"}
    a, bv, xv = rs.rand(11, 4), rs.rand(11), rs.rand(4)
    got, _, _ = run(tape_oracle, [A, b, x], [a, bv, xv], {"g": g, "g2": g2, "H": H, "H2": H2})
    want = 2 * a.T @ (a @ xv - bv) / 11
    np.testing.assert_allclose(got["g"], want, rtol=1e3 * EPS, atol=1e3 * EPS)
    np.testing.assert_allclose(got["g2"], want, rtol=1e3 * EPS, atol=1e3 * EPS)
    np.testing.assert_allclose(got["H2"], got["H"], rtol=1e3 * EPS, atol=1e3 * EPS)


def test_rewrite_leaves_constant_structure_alone(tape_oracle):
    """box constraints: F_u is built from eye/ones constants, no parameter -> nothing to hoist"""
    x = tc.Tvariable("x", (5,)); lam = tc.Tvariable("lam", (10,))
    F = tc.cat(1, x, 0.05 - x)
    Fu = tc.gradient(F, x)
    e = tc.transpose(Fu) * lam
    h = Hoister([])
    assert h.rewrite(e) is e and not h.log


def test_general_chain_and_mixed_sum(tape_oracle):
    """tprod(P, plus(fast, slow)) with a 3-factor inner product and a shared summation index"""
    rs = np.random.RandomState(4)
    P = tc.Tvariable("P", (6, 5)); Q = tc.Tvariable("Q", (5, 3)); c = tc.Tvariable("c", (5,))
    x = tc.Tvariable("x", (3,)); w = tc.Tvariable("w", (6,))
    inner = tc.tprod(Q, (1, -1), x, (-1,)) + c                 # Q x + c          (5)
    e = tc.tprod(P, (-1, 1), w, (-1,), inner, (-2,), Q, (-2, 2))   # (P'w)_i * sum_k inner_k Q_kj   (6x5)(5x3) -> [5,3]
    h = Hoister([P, Q, c])
    e2 = h.rewrite(e)
    p, q, cv, xv, wv = rs.rand(6, 5), rs.rand(5, 3), rs.rand(5), rs.rand(3), rs.rand(6)
    got, _, _ = run(tape_oracle, [P, Q, c, x, w], [p, q, cv, xv, wv], {"e": e, "e2": e2})
    want = np.outer(p.T @ wv, (q @ xv + cv) @ q)
    np.testing.assert_allclose(got["e"], want, rtol=1e3 * EPS, atol=1e3 * EPS)
    np.testing.assert_allclose(got["e2"], want, rtol=1e3 * EPS, atol=1e3 * EPS)
    assert h.fast_cost(e2) <= h.fast_cost(e)


def test_cost_is_expanded_into_a_quadratic_form(tape_oracle):
    """R3: norm2(A*x-b) -> x'(A'A)x - 2(A'b)'x + b'b; the x-dependent part no longer touches A or b, the
    value agrees to ~1e-13 relative (a difference of larger numbers instead of a sum of squares)"""
    rs = np.random.RandomState(6)
    A = tc.Tvariable("A", (100, 8)); b = tc.Tvariable("b", (100,)); x = tc.Tvariable("x", (8,))
    J = tc.norm2(A * x - b) / 100
    h = Hoister([A, b])
    J2 = h.rewrite(J)
    assert any(k == "expand norm2" for k, *_ in h.log)
    assert h.fast_cost(J2) * 4 < h.fast_cost(J)
    worst = 0.0
    for trial in range(5):
        a, bv, xv = rs.rand(100, 8), rs.rand(100), .05 * rs.rand(8)
        got, _, _ = run(tape_oracle, [A, b, x], [a, bv, xv], {"J": J, "J2": J2})
        want = ((a @ xv - bv) ** 2).sum() / 100
        np.testing.assert_allclose(got["J"], want, rtol=1e-14)
        np.testing.assert_allclose(got["J2"], want, rtol=2e-12)
        worst = max(worst, abs(float(got["J2"]) - want) / want)
    assert worst < 1e-12


def test_hoist_is_a_no_op_where_there_is_nothing_to_hoist(native_built, tmp_path):
    """mpc_dcmotor has no product of parameter arrays with the optimisation variables: identical tape"""
    from tenscalc_amd.problems.mpc_dcmotor import mpc_dcmotor
    a = mpc_dcmotor(folder=str(tmp_path), classname="tmp_dc_a").tape
    b = mpc_dcmotor(folder=str(tmp_path), classname="tmp_dc_b", hoist=True).tape
    assert np.array_equal(a.types, b.types) and np.array_equal(a.ops, b.ops) and np.array_equal(a.pars, b.pars)


@pytest.mark.parametrize("kind", ["box", "eq", "eqonly", "free"])
def test_hoisted_solver_agrees_with_the_reference_tape(native_built, tmp_path, kind):
    """whole solves through the oracle, reference tape vs hoisted tape, on the small problems that cover the
    template branches (equality constraints: second LDL and nu; nF==0; unconstrained)"""
    from tenscalc_amd.tapeio import save_tape
    from oracle.tc_oracle import Oracle
    from test_solver_options import _problem
    rs = np.random.RandomState(8)
    B = 40
    A = rs.random_sample((B, 12, 4)); b = rs.random_sample((B, 12)); x0 = .02 * rs.random_sample((B, 4)) + .005
    At = np.ascontiguousarray(np.transpose(A, (0, 2, 1))).reshape(B, -1)
    res = []
    for hoist in (False, True):
        spec = _problem(kind, hoist=hoist)
        tape = str(tmp_path / ("t%d.tape" % hoist))
        save_tape(spec.tape, tape)
        c = spec.classname
        res.append(Oracle(tape).solve_batch({c + "_set_A": At, c + "_set_b": b, c + "_set_x": x0}, c + "_getOutputs", B,
                                            mu0=1e-2, maxIter=40))
    (o0, s0, i0), (o1, s1, i1) = res
    assert np.array_equal(s0, s1) and np.abs(i0 - i1).max() <= 2
    ok = s0 == 0
    assert ok.mean() > 0.9
    assert np.abs(o0[ok] - o1[ok]).max() <= 1e-4 and np.quantile(np.abs(o0[ok] - o1[ok]).max(axis=1), 0.8) <= 1e-7


def _random_expr(rs, slow, fast, depth):
    """random scalar/vector/matrix expression over slow (parameter) and fast variables built from plus, tprod
    (matrix-vector, matrix-matrix, inner, outer, Hadamard) and norm2"""
    leaves = slow + fast
    if depth == 0:
        return leaves[rs.randint(len(leaves))]
    a = _random_expr(rs, slow, fast, depth - 1)
    b = _random_expr(rs, slow, fast, depth - 1)
    for _ in range(8):
        kind = rs.randint(6)
        try:
            if kind == 0 and a.size == b.size:
                return tc.plus([a, b], [1, -1 if rs.rand() < .5 else 1])
            if kind == 1 and a.ndim == 2 and b.ndim >= 1 and a.size[1] == b.size[0]:
                return a * b
            if kind == 2 and a.ndim == 2 and b.ndim == 1 and a.size[0] == b.size[0]:
                return tc.tprod(a, (-1, 1), b, (-1,))                    # a' b
            if kind == 3 and a.size == b.size and a.ndim >= 1:
                return tc.times(a, b)
            if kind == 4 and a.ndim == 1 and b.ndim == 1 and a.size == b.size:
                return tc.tprod(a, (-1,), b, (-1,)) * a                   # (a.b) a
            if kind == 5 and a.ndim >= 1:
                return tc.norm2(a) * b if rs.rand() < .5 else tc.plus([tc.norm2(a) * b, b], [1, 1])
        except ValueError:
            pass
        a, b = b, a
    return tc.plus([a, a], [1, 1]) if rs.rand() < .5 else a


@pytest.mark.parametrize("seed", range(12))
def test_random_expressions_evaluate_the_same_after_rewriting(tape_oracle, seed):
    """index bookkeeping of R1-R3 on random expression DAGs: value (and value of the gradient with respect to a
    fast variable, where it exists) unchanged to rounding"""
    rs = np.random.RandomState(100 + seed)
    n, m = 4, 3
    P = tc.Tvariable("P", (n, m)); Q = tc.Tvariable("Q", (m, m)); c = tc.Tvariable("c", (m,)); d = tc.Tvariable("d", (n,))
    x = tc.Tvariable("x", (m,)); y = tc.Tvariable("y", (n,))
    slow, fast = [P, Q, c, d], [x, y]
    vals = [rs.rand(n, m), rs.rand(m, m), rs.rand(m), rs.rand(n), rs.rand(m), rs.rand(n)]
    outs, nrew = {}, 0
    for k in range(6):
        e = _random_expr(rs, slow, fast, 3)
        h = Hoister(slow)
        e2 = h.rewrite(e)
        nrew += len(h.log)
        outs[f"e{k}"], outs[f"r{k}"] = e, e2
        if e.size == ():
            try:
                g_ = tc.gradient(e, x)
            except NotImplementedError:
                continue
            outs[f"g{k}"], outs[f"s{k}"] = g_, Hoister(slow).rewrite(g_)
    got, _, _ = run(tape_oracle, slow + fast, vals, outs)
    for k in range(6):
        scale = max(1.0, float(np.abs(got[f"e{k}"]).max()))
        np.testing.assert_allclose(got[f"r{k}"], got[f"e{k}"], rtol=1e-11, atol=1e-11 * scale, err_msg=f"e{k}")
        if f"g{k}" in got:
            scale = max(1.0, float(np.abs(got[f"g{k}"]).max()))
            np.testing.assert_allclose(got[f"s{k}"], got[f"g{k}"], rtol=1e-11, atol=1e-11 * scale, err_msg=f"g{k}")
