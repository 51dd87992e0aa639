"""Two-player equilibrium solvers: cmex2equilibriumLatentCS + the ipmPDeqlat_CS assembly + the
symbolic LU of @csparse/sparsity_lu.m + the Newton loop lib/ipmPDeq_CSsolver.c.

CPU: (1) the symbolic LU and its triangular solves against numpy, in the style of the reference's
examples/testing/testFactorizations.m; (2) the three loops -- the reference's own file compiled in place, the oracle's
restatement, the DEVICE loop compiled for the host -- bit-identical on the three games of examples/TCgames.m over >= 100
seeds and over the compile-time branches (addEye2Hessian adaptation, uncoupled step lengths, Mehrotra predictor step,
the smaller Newton matrix, scaleCost, LDL of the symmetrised system); (3) the computed equilibria against what the
games' own mathematics says (linear programme for the matrix game, closed form for the quadratic saddle points).
GPU (-m gpu): the generated classes, through the C-ABI, bit-identical to the oracle."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_symbolic_lu_and_solves_against_numpy(native_built, tape_oracle):
    from tenscalc_amd import tcalculus as tc
    from tenscalc_amd.csparse import CSparse
    for seed in range(6):
        tc.clear()
        n = 9 + 3 * seed
        rs = np.random.RandomState(seed)
        M = (rs.random_sample((n, n)) < .25) * rs.standard_normal((n, n))
        M[np.arange(n), np.arange(n)] = 0                    # zeros on the diagonal: row pivoting is needed
        M[np.arange(n), rs.permutation(n)] = 2 + rs.random_sample(n)
        A, b = tc.Tvariable("A", (n, n)), tc.Tvariable("b", (n, 2))
        F = tc.lu(tc.times(tc.Tconstant((M != 0).astype(float)), A))
        cs = CSparse()
        cs.declareSet(A, "set_A", external=True)
        cs.declareSet(b, "set_b", external=True)
        cs.declareGet([tc.full(tc.mldivide(F, b)), tc.full(F)], "get", ["x", "LU"])
        o = tape_oracle(cs.compile({}), "lu")
        bv = rs.standard_normal((n, 2))
        o.set("set_A", M)
        o.set("set_b", bv)
        out = o.get("get")
        x, LU = out[:2 * n].reshape((n, 2), order="F"), out[2 * n:].reshape((n, n), order="F")
        p, q = list(cs.lu_perm.values())[0]
        assert sorted(p) == list(range(n)) and sorted(q) == list(range(n))
        L, U = np.tril(LU, -1) + np.eye(n), np.triu(LU)
        assert np.abs(L @ U - M[p][:, q]).max() <= 1e3 * np.finfo(float).eps * max(1, np.abs(M).max())     # testFactorizations.m:74-90
        assert np.abs(x - np.linalg.solve(M, bv)).max() <= 1e-10


def _spec(name, **opt):
    import __graft_entry__ as g
    from tenscalc_amd.problems.tc_games import GAMES
    mk, _, kw = GAMES[name]
    kw = dict(kw)
    if opt:
        kw["classname"] = kw["classname"] + "_" + "_".join("%s%s" % (k[:6], str(v)[:3]) for k, v in sorted(opt.items()))
    return mk(folder=g.GENERATED, **kw, **opt)


def _three_loops(spec, tag, tmp_path):
    from tenscalc_amd.tapeio import save_tape
    from oracle.tc_oracle import DeviceLoopOracle, Oracle, ReferenceLoopOracle
    tape = str(tmp_path / (tag + ".tape"))
    save_tape(spec.tape, tape)
    o = Oracle(tape)
    assert o.eq
    try:
        r = ReferenceLoopOracle(tape, "eq_" + tag, spec.defines, solver="ipmPDeq_CSsolver")
    except FileNotFoundError:
        pytest.skip("needs /root/reference (or a prebuilt oracle/_ref library)")
    d = DeviceLoopOracle(tape, "eq_" + tag, spec.defines, solver="ipmPDeq_CSsolver")
    return o, r, d


def _solve(X, spec, data, args):
    c = spec.classname
    for n, _ in spec.parameters + spec.variables:
        X.set(f"{c}_set_{n}", data[n])
    st, it = X.solve(**args)
    return st, it, X.get(c + "_getOutputs")


@pytest.mark.parametrize("name", ["ms", "qs", "qss"])
def test_three_loops_bit_identical_on_the_games(native_built, tmp_path, name):
    from tenscalc_amd.problems.tc_games import SOLVE_ARGS, game_data
    spec = _spec(name)
    o, r, d = _three_loops(spec, name, tmp_path)
    its = []
    for seed in range(40):
        data = game_data(name, seed)
        if seed % 10 == 9:              # maxIter exhaustion: the epilogue's status bits
            args = dict(SOLVE_ARGS, maxIter=3 + seed % 4)
        else:
            args = SOLVE_ARGS
        res = [_solve(X, spec, data, args) for X in (r, o, d)]
        for x in res[1:]:
            assert res[0][0] == x[0] and res[0][1] == x[1] and np.array_equal(res[0][2], x[2], equal_nan=True), (name, seed)
        its.append((res[0][0], res[0][1]))
    assert sum(1 for st, _ in its if st == 0) >= 36 and any(st & 8 for st, _ in its)


@pytest.mark.parametrize("name,opt", [
    ("ms", dict(addEye2Hessian=True, adjustAddEye2Hessian=True, smallerNewtonMatrix=True)),   # Hessian2 adaptation (:321-378)
    ("ms", dict(coupledAlphas=False)),
    ("ms", dict(skipAffine=False)),                                                          # Mehrotra predictor (:419-510)
    ("qs", dict(skipAffine=False, smallerNewtonMatrix=True)),
    ("qss", dict(smallerNewtonMatrix=True)),
    ("qs", dict(scaleCost=10.0)),
    ("qs", dict(useLDL=True)),                                                               # symmetrised system (:645-651)
    ("qss", dict(skipAffine=False, delta=2)),
])
def test_three_loops_bit_identical_on_template_branches(native_built, tmp_path, name, opt):
    from tenscalc_amd.problems.tc_games import SOLVE_ARGS, game_data
    spec = _spec(name, **opt)
    tag = name + "_" + "_".join(sorted(opt))[:40]
    o, r, d = _three_loops(spec, tag, tmp_path)
    ok = 0
    for seed in range(15):
        data = game_data(name, 100 + seed)
        if seed == 3:                   # a NaN parameter / start: status 4
            data = dict(data)
            k0 = list(data)[0]
            v = np.array(data[k0], dtype=float)
            v.flat[0] = np.nan
            data[k0] = v
        args = dict(SOLVE_ARGS, addEye2Hessian=(1e-9, 1e-7))
        res = [_solve(X, spec, data, args) for X in (r, o, d)]
        for x in res[1:]:
            assert res[0][0] == x[0] and res[0][1] == x[1] and np.array_equal(res[0][2], x[2], equal_nan=True), (name, opt, seed)
        ok += res[0][0] == 0
    assert ok >= 8, (name, opt, ok)


def test_matrix_game_value_against_linear_programming(native_built, tmp_path):
    """the mixed-policy saddle point of u'*A*d (player 1 minimises) is the solution of a linear programme"""
    from scipy.optimize import linprog
    from tenscalc_amd.problems.tc_games import SOLVE_ARGS, game_data
    from tenscalc_amd.tapeio import save_tape
    from oracle.tc_oracle import Oracle
    spec = _spec("ms")
    tape = str(tmp_path / "ms.tape")
    save_tape(spec.tape, tape)
    o = Oracle(tape)
    for seed in range(5):
        data = game_data("ms", 300 + seed)
        st, it, out = _solve(o, spec, data, SOLVE_ARGS)
        A = data["A1"]
        N1, N2 = A.shape
        u, d, J = out[:N1], out[N1:N1 + N2], out[-1]
        # min_u max_j (A'u)_j: variables (u, v), minimise v s.t. A'u - v <= 0, sum u = 1, u >= 0
        lp = linprog(np.r_[np.zeros(N1), 1.0], A_ub=np.c_[A.T, -np.ones(N2)], b_ub=np.zeros(N2),
                     A_eq=np.r_[np.ones(N1), 0.0][None], b_eq=[1.0], bounds=[(0, None)] * N1 + [(None, None)])
        assert st == 0 and lp.status == 0
        assert abs(u.sum() - 1) <= 1e-4 and abs(d.sum() - 1) <= 1e-4 and u.min() >= 0 and d.min() >= 0
        assert abs(J - lp.fun) <= 2e-4, (J, lp.fun)            # desiredDualityGap 1e-5, gradTolerance 1e-4
        assert (A.T @ u).max() <= lp.fun + 1e-3 and (A @ d).min() >= lp.fun - 1e-3     # neither player can improve


def test_predictor_step_on_the_symmetrised_system(native_built, tmp_path):
    """skipAffine=false together with useLDL=true: the reference rebuilds the combined right-hand side after the predictor
    step without the WW' factor of the symmetrised system (ipmPDeqlat_CS.m:646-648 vs :702-707 large matrix; :419-421 vs
    :473-479 small matrix).  Here the LARGE matrix keeps the factor (a documented deviation, equilibrium.py): the solver
    converges to the closed form, as with LU; the SMALL matrix follows the reference's formula and, like it, does not."""
    from tenscalc_amd.problems.tc_games import GAMES, SOLVE_ARGS, game_data
    from tenscalc_amd.tapeio import save_tape
    from oracle.tc_oracle import Oracle
    N = GAMES["qs"][2]["N"]
    sol = {}
    for tag, opt in (("lu", dict(skipAffine=False)), ("ldl", dict(skipAffine=False, useLDL=True)),
                     ("ldl_small", dict(skipAffine=False, useLDL=True, smallerNewtonMatrix=True))):
        spec = _spec("qs", **opt)
        tape = str(tmp_path / (tag + ".tape"))
        save_tape(spec.tape, tape)
        o = Oracle(tape)
        sol[tag] = [_solve(o, spec, game_data("qs", 100 + seed), SOLVE_ARGS) for seed in range(4)]
    for (st, it, out), (st2, it2, out2) in zip(sol["lu"], sol["ldl"]):
        assert st == 0 and st2 == 0 and abs(it - it2) <= 1
        assert np.abs(out2[:N] - 1).max() <= 1e-4 and np.abs(out2[N:2 * N] + 1 / 9).max() <= 1e-4
        assert np.abs(out - out2).max() <= 1e-5
    assert all(st & 8 for st, _, _ in sol["ldl_small"])          # the reference's formula: maxIter, large gradient


@pytest.mark.parametrize("name", ["qs", "qss"])
def test_quadratic_saddle_point_closed_form(native_built, tmp_path, name):
    """Ju = |u+d-2|^2 - 10|d|^2 on the box [-1,1]^N: player 1 sits at u = 1, player 2's stationarity
    2(u+d-2) - 20 d = 0 gives d = -1/9, Ju = N*10/9 (with the latent variable: x = u + d = 8/9)"""
    from tenscalc_amd.problems.tc_games import GAMES, SOLVE_ARGS, game_data
    from tenscalc_amd.tapeio import save_tape
    from oracle.tc_oracle import Oracle
    spec = _spec(name)
    N = GAMES[name][2]["N"]
    tape = str(tmp_path / "q.tape")
    save_tape(spec.tape, tape)
    o = Oracle(tape)
    st, it, out = _solve(o, spec, game_data(name, 5), SOLVE_ARGS)
    assert st == 0
    u, d = out[:N], out[N:2 * N]
    assert np.abs(u - 1).max() <= 1e-4 and np.abs(d + 1 / 9).max() <= 1e-4
    if name == "qss":
        assert np.abs(out[2 * N:3 * N] - 8 / 9).max() <= 1e-4
        assert abs(out[3 * N] - N * 10 / 9) <= 1e-3 and out[3 * N] == out[3 * N + 1]
    else:
        assert abs(out[2 * N] - N * 10 / 9) <= 1e-3


def test_typical_values_files_order_the_lu(native_built, tmp_path, tape_oracle):
    """allowSave workflow (sparsity_lu.m:36-57, writeCfunctionpergroup.m:879-918): generation 1 writes <cls>_WW.subscripts and
    orders the LU with random values; a saved <cls>_WW.values (here produced through the oracle's saveWW__ get; on the GPU
    by HipSolver.saveWW) makes generation 2 order it with the typical matrix; the solution does not change beyond
    rounding; a pair with mismatching magic numbers is ignored (loadCSparse.m)."""
    from tenscalc_amd import typical
    from tenscalc_amd.problems import tc_games as tg
    kw = dict(N1=7, N2=5, classname="tmp_ms_save", folder=str(tmp_path), allowSave=True)
    spec = tg.matrixsaddle(**kw)
    base = str(tmp_path / "@tmp_ms_save" / "tmp_ms_save_WW")
    assert spec.code.lu_stats[-1]["order_from"] == "random values" and spec.tape.has("saveWW__")
    magic, osize, subs = typical.read_subscripts(base + ".subscripts")
    n = 7 + 5 + 2 + 12
    assert osize == (n, n) and subs.min() >= 0 and subs.max() < n
    d = tg.matrixsaddle_data(3, 7, 5)
    o = tape_oracle(spec.tape, "save")
    c = spec.classname
    for nm, _ in spec.parameters + spec.variables:
        o.set(f"{c}_set_{nm}", d[nm])
    for e in ("initPrimal__", "scaleIneq__"):
        o.copy(e)
    o.set("setMu__", [1.0])
    o.copy("initDualIneq__")
    o.copy("initDualEq__")
    W = o.get("saveWW__").reshape(osize, order="F")
    assert np.count_nonzero(W) <= len(subs) and np.count_nonzero(W[subs[:, 0], subs[:, 1]]) == np.count_nonzero(W)
    typical.write_values(base + ".values", W[subs[:, 0], subs[:, 1]], magic)
    r = typical.load(base + ".subscripts", base + ".values")
    assert r is not None and r[0] == osize and np.array_equal(r[1], subs)
    st1, it1, y1 = _solve(o, spec, d, tg.SOLVE_ARGS)
    spec2 = tg.matrixsaddle(**kw)                               # generation 2: reads the pair, then writes new subscripts
    assert spec2.code.lu_stats[-1]["order_from"] == "typical values"
    o2 = tape_oracle(spec2.tape, "save2")
    st2, it2, y2 = _solve(o2, spec2, d, tg.SOLVE_ARGS)
    assert st1 == st2 == 0 and abs(it1 - it2) <= 1 and np.abs(y1 - y2).max() <= 1e-6
    assert typical.load(base + ".subscripts", base + ".values") is None           # new magic: the old values no longer match
    spec3 = tg.matrixsaddle(**kw)
    assert spec3.code.lu_stats[-1]["order_from"] == "random values"


def test_front_door_argument_checks(native_built):
    from tenscalc_amd import tcalculus as tc
    from tenscalc_amd.equilibrium import cmex2equilibriumLatentCS
    tc.clear()
    u, d, x = tc.Tvariable("u", (3,)), tc.Tvariable("d", (3,)), tc.Tvariable("x", (3,))
    J = tc.norm2(u + d)
    with pytest.raises(ValueError, match="scalar"):
        cmex2equilibriumLatentCS(classname="t", P1objective=u, P2objective=-J, P1optimizationVariables=[u],
                                 P2optimizationVariables=[d])
    with pytest.raises(ValueError, match="latent constraints cannot be inequalities"):
        cmex2equilibriumLatentCS(classname="t", P1objective=J, P2objective=-J, P1optimizationVariables=[u],
                                 P2optimizationVariables=[d], latentVariables=[x], latentConstraints=[x >= 0])
    with pytest.raises(ValueError, match="must match number of latent"):
        cmex2equilibriumLatentCS(classname="t", P1objective=J, P2objective=-J, P1optimizationVariables=[u],
                                 P2optimizationVariables=[d], latentVariables=[x])
    with pytest.raises(ValueError, match="cannot also be a parameter"):
        cmex2equilibriumLatentCS(classname="t", P1objective=J, P2objective=-J, P1optimizationVariables=[u],
                                 P2optimizationVariables=[d], parameters=[d])
    with pytest.raises(TypeError, match="unknown option"):
        cmex2equilibriumLatentCS(classname="t", P1objective=J, P2objective=-J, P1optimizationVariables=[u],
                                 P2optimizationVariables=[d], gradTolerence=1e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["ms", "qs", "qss"])
@pytest.mark.parametrize("variant", ["strict", "interp"])
def test_gpu_equilibrium_classes_bit_exact(tmp_path, name, variant):
    """the generated class through the C-ABI (tc_create / tc_set / tc_solve / tc_get_outputs) against the oracle: status,
    iteration count and every output bit for bit, incl. instances stopped by maxIter and a NaN instance"""
    import __graft_entry__ as g
    from tenscalc_amd.problems.tc_games import SOLVE_ARGS, game_data
    from tenscalc_amd.runtime import HipSolver
    from tenscalc_amd.tapeio import save_tape
    from oracle.tc_oracle import Oracle
    spec = _spec(name)
    so = spec.build(**g.VARIANT_KW[variant])
    B = 150
    data = [game_data(name, 1000 + k) for k in range(B)]
    names = [n for n, _ in spec.parameters + spec.variables]
    stacked = {n: np.stack([np.asarray(d[n], dtype=float).reshape(-1, order="F") for d in data]) for n in names}
    stacked[names[0]][7, 0] = np.nan
    tape = str(tmp_path / "g.tape")
    save_tape(spec.tape, tape)
    c = spec.classname
    for maxIter in (SOLVE_ARGS["maxIter"], 4):
        obj = HipSolver(so, B=B, device=0)
        for n in names:
            obj.set(n, stacked[n])
        st, it, _ = obj.solve(**dict(SOLVE_ARGS, maxIter=maxIter))
        got = np.concatenate([a.reshape(B, -1) for a in obj.getOutputs()], axis=1)
        obj.close()
        ref, rst, rit = Oracle(tape).solve_batch({f"{c}_set_{n}": stacked[n] for n in names}, c + "_getOutputs", B,
                                                 mu0=SOLVE_ARGS["mu0"], maxIter=maxIter)
        assert np.array_equal(st, rst) and np.array_equal(it, rit), (name, variant, maxIter)
        assert np.array_equal(got, ref, equal_nan=True)
        if maxIter > 4:
            assert (st == 0).sum() >= B - 3 and st[7] == 4


@pytest.mark.gpu
def test_gpu_saveWW_writes_the_reference_file_pair(tmp_path):
    """HipSolver.saveWW: the Newton matrix of one instance after the initialisation sequence, in the reference's
    .subscripts / .values format, equal to the oracle's saveWW__ at the same point"""
    import __graft_entry__ as g
    from tenscalc_amd import typical
    from tenscalc_amd.problems import tc_games as tg
    from tenscalc_amd.runtime import HipSolver
    from tenscalc_amd.tapeio import save_tape
    from oracle.tc_oracle import Oracle
    spec = tg.matrixsaddle(N1=7, N2=5, classname="tmp_ms_save", folder=str(tmp_path), allowSave=True)
    so = spec.build(**g.VARIANT_KW["strict"])
    B = 5
    data = [tg.matrixsaddle_data(40 + k, 7, 5) for k in range(B)]
    obj = HipSolver(so, B=B, device=0)
    for n, _ in spec.parameters + spec.variables:
        obj.set(n, np.stack([np.asarray(d[n], dtype=float).reshape(-1, order="F") for d in data]))
    path = obj.saveWW(instance=3)
    obj.close()
    base = path[:-len(".values")]
    osize, subs, vals = typical.load(base + ".subscripts", path)
    tape = str(tmp_path / "s.tape")
    save_tape(spec.tape, tape)
    o = Oracle(tape)
    c = spec.classname
    for n, _ in spec.parameters + spec.variables:
        o.set(f"{c}_set_{n}", data[3][n])
    for e in ("initPrimal__", "scaleIneq__"):
        o.copy(e)
    o.set("setMu__", [1.0])
    o.copy("initDualIneq__")
    o.copy("initDualEq__")
    W = o.get("saveWW__").reshape(osize, order="F")
    assert np.array_equal(vals, W[subs[:, 0], subs[:, 1]])
    spec2 = tg.matrixsaddle(N1=7, N2=5, classname="tmp_ms_save", folder=str(tmp_path), allowSave=True)
    assert spec2.code.lu_stats[-1]["order_from"] == "typical values"
