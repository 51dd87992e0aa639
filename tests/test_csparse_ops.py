"""Scalarizers checked through the CPU oracle against numpy dense algebra -- the strategy of the
reference's own unit tests (examples/testing/testLinearAlgebra.m:118-190 tol 1e3*eps,
testNorms.m:85-117, testComponentwise.m:98-108, testFactorizations.m:74-90, testLDL.m:84-92)."""
import numpy as np
import pytest

import tenscalc_amd.tcalculus as tc
from tenscalc_amd.cmex2compute import compute_tape
from tenscalc_amd.csparse import min_degree_order

EPS = np.finfo(float).eps


def run(tape_oracle, inputs, values, outputs):
    tape, code = compute_tape(inputs, outputs)
    o = tape_oracle(tape)
    for v, val in zip(inputs, values):
        o.set(f"tmpC_set_{v.par}", np.asarray(val, dtype=float))
    res = {}
    for nm, e in outputs.items():
        res[nm] = o.get(f"tmpC_get_{nm}").reshape(e.size, order="F")
    return res, tape, code


def test_linear_algebra(tape_oracle):
    rs = np.random.RandomState(0)
    A = tc.Tvariable("A", (5, 4)); B = tc.Tvariable("B", (4, 3)); x = tc.Tvariable("x", (4,))
    y = tc.Tvariable("y", (5,)); s = tc.Tvariable("s", ())
    a, b, xv, yv, sv = rs.rand(5, 4), rs.rand(4, 3), rs.rand(4), rs.rand(5), rs.rand()
    outs = {
        "AB": A * B, "Ax": A * x, "yA": y * A, "xx": x * x, "sA": s * A, "As": A * s,
        "AtA": A.T * A, "plus": A * x - y + 2 * y, "neg": -(A * x), "tp": tc.tprod(A, (1, -1), B, (-1, 2), x, (-1,)),
        "outer": tc.tprod(y, (1,), x, (2,)), "trace": tc.tprod(A.T * A, (-1, -1)),
        "sum1": tc.tsum(A, 1), "sumall": tc.tsum(A), "times": tc.times(A * x, y), "div": (A * x) / y,
        "divs": A / s, "sdiv": s / y, "eyeprod": tc.Teye((4, 4)) * x, "ones": tc.Tones((5,)) * s + y,
    }
    want = {
        "AB": a @ b, "Ax": a @ xv, "yA": yv @ a, "xx": xv @ xv, "sA": sv * a, "As": a * sv,
        "AtA": a.T @ a, "plus": a @ xv - yv + 2 * yv, "neg": -(a @ xv), "tp": np.einsum("ik,kj,k->ij", a, b, xv),
        "outer": np.outer(yv, xv), "trace": np.trace(a.T @ a), "sum1": a.sum(0), "sumall": a.sum(),
        "times": (a @ xv) * yv, "div": (a @ xv) / yv, "divs": a / sv, "sdiv": sv / yv, "eyeprod": xv,
        "ones": sv + yv,
    }
    got, _, _ = run(tape_oracle, [A, B, x, y, s], [a, b, xv, yv, sv], outs)
    for k in want:
        np.testing.assert_allclose(got[k], want[k], rtol=1e3 * EPS, atol=1e3 * EPS, err_msg=k)


def test_structural_ops_emit_no_instructions(tape_oracle):
    rs = np.random.RandomState(1)
    A = tc.Tvariable("A", (3, 4)); v = tc.Tvariable("v", (4,))
    a, vv = rs.rand(3, 4), rs.rand(4)
    outs = {
        "cat1": tc.cat(1, A, tc.reshape(v, (1, 4))), "cat2": tc.cat(2, A, tc.Tones((3, 2))),
        "sub": A[1:3, [0, 3]], "row": A[2, :], "resh": tc.reshape(A, (4, 3)), "tr": A.T,
        "diagm": tc.full(tc.diag(v)), "diagv": tc.diag(tc.diag(v)), "rep": tc.repmat(v, (2,)),
        "fullz": tc.full(tc.cat(1, v, tc.Tzeros((2,)))),
    }
    want = {
        "cat1": np.vstack([a, vv[None]]), "cat2": np.hstack([a, np.ones((3, 2))]),
        "sub": a[1:3][:, [0, 3]], "row": a[2], "resh": a.reshape((4, 3), order="F"), "tr": a.T,
        "diagm": np.diag(vv), "diagv": vv, "rep": np.tile(vv, 2), "fullz": np.r_[vv, 0, 0],
    }
    got, tape, _ = run(tape_oracle, [A, v], [a, vv], outs)
    for k in want:
        np.testing.assert_array_equal(got[k], want[k], err_msg=k)
    # only I_set and I_load: cat/subsref/reshape/transpose/diag are pure aliasing (sparsity_cat.m,...)
    assert set(np.unique(tape.types)) <= {1, 2}


def test_norms_minmax_clp_compose(tape_oracle):
    rs = np.random.RandomState(2)
    x = tc.Tvariable("x", (6,)); d = tc.Tvariable("d", (6,)); M = tc.Tvariable("M", (3, 4))
    xv, dv, m = rs.rand(6) + .1, rs.randn(6), rs.randn(3, 4)
    outs = {
        "n2": tc.norm2(x - d), "ninf": tc.norminf(d), "n1": tc.norm1(d), "min": tc.tmin(d), "max": tc.tmax(d),
        "mincol": tc.tmin(M, 1), "maxrow": tc.tmax(M, 2), "clp": tc.clp(x, d), "min2": tc.min2(x, d),
        "max2": tc.max2(x, 0.5), "abs": tc.tabs(d), "sqrt": tc.sqrt(x), "sqr": tc.sqr(d), "cube": tc.cube(d),
        "exp": tc.exp(d), "log": tc.log(x), "sin": tc.sin(d), "cos": tc.cos(d), "inv": 1 / x,
        "heav": tc.heaviside(d), "relu": tc.relu(d),
    }
    neg = dv < 0
    want = {
        "n2": ((xv - dv) ** 2).sum(), "ninf": np.abs(dv).max(), "n1": np.abs(dv).sum(), "min": dv.min(),
        "max": dv.max(), "mincol": m.min(0), "maxrow": m.max(1),
        "clp": (-xv[neg] / dv[neg]).min() if neg.any() else np.finfo(float).max,
        "min2": np.minimum(xv, dv), "max2": np.maximum(xv, .5), "abs": np.abs(dv), "sqrt": np.sqrt(xv),
        "sqr": dv ** 2, "cube": dv ** 3, "exp": np.exp(dv), "log": np.log(xv), "sin": np.sin(dv),
        "cos": np.cos(dv), "inv": 1 / xv, "heav": (dv > 0) * 1.0, "relu": np.maximum(dv, 0),
    }
    got, _, _ = run(tape_oracle, [x, d, M], [xv, dv, m], outs)
    for k in want:
        np.testing.assert_allclose(got[k], want[k], rtol=10 * EPS, atol=10 * EPS, err_msg=k)


def test_gradient_hessian_finite_differences(tape_oracle):
    rs = np.random.RandomState(3)
    A = tc.Tvariable("A", (5, 3)); b = tc.Tvariable("b", (5,)); u = tc.Tvariable("u", (3,))
    a, bv, uv = rs.rand(5, 3), rs.rand(5), rs.rand(3) + .5
    f = tc.norm2(A * u - b) / 5 + tc.tsum(tc.log(u)) + tc.tsum(tc.times(tc.sqrt(u), tc.exp(-u))) \
        + (u * u) / (1 + u[0]) + tc.tsum(tc.cube(u[0:2]))
    H, g = tc.hessian(f, u)
    got, _, _ = run(tape_oracle, [A, b, u], [a, bv, uv], {"f": f, "g": tc.full(g), "H": tc.full(H)})

    def fun(z):
        return ((a @ z - bv) ** 2).sum() / 5 + np.log(z).sum() + (np.sqrt(z) * np.exp(-z)).sum() \
            + (z @ z) / (1 + z[0]) + (z[:2] ** 3).sum()
    np.testing.assert_allclose(got["f"], fun(uv), rtol=1e-14)
    h = 1e-6
    gfd = np.array([(fun(uv + h * e) - fun(uv - h * e)) / (2 * h) for e in np.eye(3)])
    np.testing.assert_allclose(got["g"], gfd, rtol=1e-7, atol=1e-8)
    Hfd = np.zeros((3, 3))
    h = 1e-4
    for i, ei in enumerate(np.eye(3)):
        for j, ej in enumerate(np.eye(3)):
            Hfd[i, j] = (fun(uv + h * ei + h * ej) - fun(uv + h * ei - h * ej)
                         - fun(uv - h * ei + h * ej) + fun(uv - h * ei - h * ej)) / (4 * h * h)
    np.testing.assert_allclose(got["H"], Hfd, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(got["H"], got["H"].T, rtol=0, atol=1e-14)


def test_substitute(tape_oracle):
    x = tc.Tvariable("x", (3,)); y = tc.Tvariable("y", (3,)); a = tc.Tvariable("a", ())
    f = tc.tmin(tc.times(x, x) - 1)
    g = tc.substitute(f, x, x + a * y)
    xv, yv, av = np.array([1., 2., 3.]), np.array([.5, -1., 2.]), .25
    got, _, _ = run(tape_oracle, [x, y, a], [xv, yv, av], {"g": g})
    np.testing.assert_allclose(got["g"], ((xv + av * yv) ** 2 - 1).min(), rtol=1e-15)


def _random_sym(rs, n, density, shift):
    M = np.zeros((n, n))
    for i in range(n):
        for j in range(i):
            if rs.rand() < density:
                M[i, j] = M[j, i] = rs.randn()
    M += np.diag(shift + rs.rand(n))
    return M


@pytest.mark.parametrize("n,density,seed", [(6, .5, 0), (12, .3, 1), (25, .15, 2), (40, .08, 3)])
def test_ldl_factorization_and_solve(tape_oracle, n, density, seed):
    """L*diag(D)*L' == A(p,p) to 1e3*eps (testFactorizations.m:74-90) and ldl(A)\\b == A\\b
    (testLDL.m:84-92) on random sparse symmetric matrices."""
    rs = np.random.RandomState(seed)
    Mv = _random_sym(rs, n, density, n)        # diagonally dominant: no pivoting needed
    nz = np.argwhere(Mv != 0)
    vals = tc.Tvariable("vals", (len(nz),))
    # assemble the sparse symbolic matrix from its nonzeros: A = sum_k vals[k] * E_k via a constant 3-tensor
    E = np.zeros((n, n, len(nz)))
    for k, (i, j) in enumerate(nz):
        E[i, j, k] = 1.0
    Asym = tc.tprod(tc.Tconstant(E), (1, 2, -1), vals, (-1,))
    b = tc.Tvariable("b", (n,)); Bm = tc.Tvariable("Bm", (n, 2))
    F = tc.ldl(Asym)
    outs = {"x": tc.full(tc.mldivide(F, b)), "X": tc.full(tc.mldivide(F, Bm)), "d": tc.ldl_d(F),
            "L": tc.full(tc.ldl_l(F))}
    bv, Bv = rs.randn(n), rs.randn(n, 2)
    got, tape, code = run(tape_oracle, [vals, b, Bm], [Mv[nz[:, 0], nz[:, 1]], bv, Bv], outs)
    p = code.ldl_perm[F.idx]
    L, d = got["L"], got["d"]
    assert np.allclose(np.triu(L, 1), 0) and np.allclose(np.diag(L), 1)
    scale = np.abs(Mv).max()
    np.testing.assert_allclose(L @ np.diag(d) @ L.T, Mv[np.ix_(p, p)], rtol=0, atol=1e3 * EPS * scale * n)
    np.testing.assert_allclose(got["x"], np.linalg.solve(Mv, bv), rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(got["X"], np.linalg.solve(Mv, Bv), rtol=1e-11, atol=1e-12)
    st = code.ldl_stats[0]
    assert st["nnzL"] >= (len(nz) - n) // 2   # fill can only add entries


def test_ldl_quasi_definite_kkt(tape_oracle):
    """KKT-shaped matrix [H G';G -eI]: indefinite, still factorizable without pivoting."""
    rs = np.random.RandomState(5)
    n, m = 7, 3
    Hs = tc.Tvariable("H", (n, n)); G = tc.Tvariable("G", (m, n)); e = tc.Tvariable("e", ()); r = tc.Tvariable("r", (n + m,))
    K = tc.cat(1, tc.cat(2, Hs, G.T), tc.cat(2, G, -(e * tc.Teye((m, m)))))
    h = rs.rand(n, n); h = h @ h.T + np.eye(n); g = rs.randn(m, n); rv = rs.randn(n + m)
    got, _, code = run(tape_oracle, [Hs, G, e, r], [h, g, 1e-9, rv], {"x": tc.full(tc.mldivide(tc.ldl(K, stages=[0] * n + [1] * m), r))})
    Kv = np.block([[h, g.T], [g, -1e-9 * np.eye(m)]])
    np.testing.assert_allclose(got["x"], np.linalg.solve(Kv, rv), rtol=1e-8, atol=1e-9)
    assert code.ldl_stats[0]["nonsym"] > 0    # H is a free (nonsymmetric) variable here: lower entries used


def test_ldl_needs_pivoting_is_a_codegen_error(native_built):
    x = tc.Tvariable("x", (2,))
    A = tc.cat(1, tc.cat(2, tc.Tzeros((1, 1)), tc.reshape(x[0], (1, 1))), tc.cat(2, tc.reshape(x[0], (1, 1)), tc.reshape(x[1], (1, 1))))
    from tenscalc_amd.csparse import CSparse
    code = CSparse()
    with pytest.raises(ValueError, match="ldl needs pivoting"):
        code.scalarize(tc.ldl(A))


def test_min_degree_order_is_a_permutation_and_limits_fill():
    # arrow matrix: eliminating the hub last gives zero fill
    n = 30
    rows = [0] * (n - 1) + list(range(1, n)) + list(range(n))
    cols = list(range(1, n)) + [0] * (n - 1) + list(range(n))
    p = min_degree_order(n, rows, cols)
    assert sorted(p.tolist()) == list(range(n))
    assert p[-1] == 0 or p[-2] == 0


def _tridiag(n):
    rows = list(range(n)) + list(range(n - 1)) + list(range(1, n))
    cols = list(range(n)) + list(range(1, n)) + list(range(n - 1))
    return rows, cols


def test_multiple_elimination_order_reduces_a_chain_odd_even():
    """a tridiagonal matrix: front-to-back elimination has a dependency chain of n, rounds of independent
    nodes within one of the minimum degree reduce it odd-even (cyclic reduction) at twice the fill"""
    from tenscalc_amd.csparse import multiple_elimination_order, symbolic_ldl_cost
    n = 257
    rows, cols = _tridiag(n)
    p0 = min_degree_order(n, rows, cols)
    p1 = multiple_elimination_order(n, rows, cols, slack=1)      # the two ends have degree 1, the interior 2
    assert sorted(p1.tolist()) == list(range(n))
    c0, c1 = symbolic_ldl_cost(n, rows, cols, p0, term_weight=0.0), symbolic_ldl_cost(n, rows, cols, p1, term_weight=0.0)
    assert c0["nnzL"] == n - 1 and c0["depth"] >= 2 * (n - 2)        # the chain: factor + forward + backward
    assert c1["nnzL"] <= 2 * n and c1["depth"] <= 6 * np.ceil(np.log2(n))


def test_multiple_elimination_order_respects_the_gates():
    """equality duals (stage 1) only after one eliminated primal neighbour; a primal node only after ALL its
    inequality duals (stage 2)"""
    from tenscalc_amd.csparse import multiple_elimination_order
    rs = np.random.RandomState(3)
    nU, nG, nF = 20, 8, 12
    n = nU + nG + nF
    rows, cols = list(range(n)), list(range(n))
    for i in range(nU - 1):
        rows += [i, i + 1]; cols += [i + 1, i]
    for g in range(nG):
        for u in rs.choice(nU, 3, replace=False):
            rows += [nU + g, u]; cols += [u, nU + g]
    for f in range(nF):
        u = rs.randint(nU)
        rows += [nU + nG + f, u]; cols += [u, nU + nG + f]
    stages = [0] * nU + [1] * nG + [2] * nF
    adj = [set() for _ in range(n)]
    for r, c in zip(rows, cols):
        if r != c:
            adj[r].add(c)
    for slack in (0, 2, 8):
        for eq_gate, kmin in (("one", lambda k: 1), ("two", lambda k: min(2, k)), ("allbut1", lambda k: max(1, k - 1))):
            p = multiple_elimination_order(n, rows, cols, stages, slack, eq_gate=eq_gate)
            assert sorted(p.tolist()) == list(range(n))
            pos = np.empty(n, dtype=int); pos[p] = np.arange(n)
            for g in range(nU, nU + nG):
                prim = [u for u in adj[g] if u < nU]
                assert sum(pos[u] < pos[g] for u in prim) >= kmin(len(prim)), (eq_gate, g)
            for u in range(nU):
                assert all(pos[f] < pos[u] for f in adj[u] if f >= nU + nG)


@pytest.mark.parametrize("order", [("mmd", 0), ("mmd", 3), ("depth", 64)])
def test_ldl_with_parallel_orders_solves_like_numpy(tape_oracle, order):
    """the factorisation and both triangular solves under the depth-aware orders: L*diag(D)*L' == A(p,p),
    ldl(A)\\b == A\\b (testFactorizations.m:74-90, testLDL.m:84-92) on a banded matrix"""
    rs = np.random.RandomState(11)
    n = 40
    Mv = np.zeros((n, n))
    for k in (1, 2):
        v = rs.randn(n - k)
        Mv += np.diag(v, k) + np.diag(v, -k)
    Mv += np.diag(np.abs(Mv).sum(1) + 1.0)
    nz = np.argwhere(Mv != 0)
    vals = tc.Tvariable("vals", (len(nz),))
    E = np.zeros((n, n, len(nz)))
    for k, (i, j) in enumerate(nz):
        E[i, j, k] = 1.0
    Asym = tc.tprod(tc.Tconstant(E), (1, 2, -1), vals, (-1,))
    b = tc.Tvariable("b", (n,))
    F = tc.ldl(Asym)
    outs = {"x": tc.full(tc.mldivide(F, b)), "d": tc.ldl_d(F), "L": tc.full(tc.ldl_l(F))}
    bv = rs.randn(n)
    import tenscalc_amd.csparse as cs
    old = cs.LDL_ORDER
    cs.LDL_ORDER = order
    try:
        got, tape, code = run(tape_oracle, [vals, b], [Mv[nz[:, 0], nz[:, 1]], bv], outs)
    finally:
        cs.LDL_ORDER = old
    p = code.ldl_perm[F.idx]
    L, d = got["L"], got["d"]
    np.testing.assert_allclose(L @ np.diag(d) @ L.T, Mv[np.ix_(p, p)], rtol=0, atol=1e3 * EPS * np.abs(Mv).max() * n)
    np.testing.assert_allclose(got["x"], np.linalg.solve(Mv, bv), rtol=1e-11, atol=1e-12)
    assert cs.symbolic_ldl_cost(n, nz[:, 0], nz[:, 1], p)["depth"] < cs.symbolic_ldl_cost(
        n, nz[:, 0], nz[:, 1], min_degree_order(n, nz[:, 0], nz[:, 1]))["depth"]


def test_depth_aware_order_keeps_tolerance_zero_with_equality_duals():
    """with stage-1 nodes in the matrix the search stays with the greedy order or tolerance 0 (numerical finding of
    DESIGN.md 7); without them it may trade fill for depth"""
    from tenscalc_amd.csparse import depth_aware_order, multiple_elimination_order
    n = 64
    rows, cols = _tridiag(n)
    rows += [n, 0, n, 5]; cols += [0, n, 5, n]; rows.append(n); cols.append(n)
    stages = [0] * n + [1]
    p = depth_aware_order(n + 1, rows, cols, stages, width=64)
    assert (p == multiple_elimination_order(n + 1, rows, cols, stages, 0)).all() or (p == min_degree_order(n + 1, rows, cols, stages)).all()


def test_dedup_makes_symmetric_products_share_instructions(native_built):
    """A'*A: entry (i,j) and (j,i) must be the same instruction (needed by the symmetry test of
    sparsity_ldl.m:65-75; csparse.m:1168-1227 normalize)."""
    A = tc.Tvariable("A", (6, 3))
    from tenscalc_amd.csparse import CSparse
    code = CSparse()
    subs, ins = code.scalarize(A.T * A)
    M = np.zeros((3, 3), dtype=np.int64)
    M[subs[:, 0], subs[:, 1]] = ins
    assert (M == M.T).all() and len(np.unique(ins)) == 6
