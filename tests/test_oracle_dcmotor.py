"""CPU: mpc_dcmotor (T=30; equality + inequality constraints, nU=89 nG=60 nF=174) through the
oracle: structure, the reference's own Newton loop (bit-for-bit), and an independent QP solve."""
import numpy as np
import pytest

from tenscalc_amd.problems.mpc_dcmotor import PARAMS, SOLVE_ARGS, dcmotor_data


@pytest.fixture(scope="module")
def dc(native_built, tmp_path_factory):
    import __graft_entry__ as g
    from tenscalc_amd.problems.mpc_dcmotor import mpc_dcmotor
    from tenscalc_amd.tapeio import save_tape
    spec = mpc_dcmotor(folder=g.GENERATED)
    p = str(tmp_path_factory.mktemp("tapes") / "dc.tape")
    save_tape(spec.tape, p)
    return spec, p


def _set_all(q, spec, d):
    c = spec.classname
    for nm, _ in spec.parameters + spec.variables:
        q.set(f"{c}_set_{nm}", d[nm])


def test_sizes_match_survey(dc):
    spec, _ = dc
    d = spec.defines
    assert (d["nU"], d["nG"], d["nF"]) == (89, 60, 174)          # KKT 323x323
    kkt = [s for s in spec.tape.ldl_stats if s["tag"].endswith("_WW")][0]
    assert kkt["n"] == 323 and kkt["nonsym"] == 0
    names = {e["name"] for e in spec.tape.copies + spec.tape.gets}
    assert {"initDualEqX__", "initDualEq__", "getNorminf_G__"} <= names


def test_reference_newton_loop_bitwise(dc):
    from oracle.tc_oracle import Oracle, ReferenceLoopOracle
    spec, tape = dc
    try:
        ro = ReferenceLoopOracle(tape, "dcmotor", spec.defines)
    except (FileNotFoundError, OSError):
        pytest.skip("oracle/_ref not available")
    o = Oracle(tape)
    c = spec.classname
    for seed in range(12):
        d = dcmotor_data(seed)
        r = []
        for q in (o, ro):
            _set_all(q, spec, d)
            st, it = q.solve(SOLVE_ARGS["mu0"], SOLVE_ARGS["maxIter"], -1, SOLVE_ARGS["addEye2Hessian"])
            r.append((st, it, q.get(c + "_getOutputs")))
        assert r[0][:2] == r[1][:2] and r[0][0] == 0
        assert np.array_equal(r[0][2], r[1][2])


def test_against_independent_qp_solver(dc):
    """the MPC problem is a convex QP: compare with scipy SLSQP on the same data"""
    from scipy.optimize import minimize
    from oracle.tc_oracle import Oracle
    spec, tape = dc
    o = Oracle(tape)
    c = spec.classname
    T, Ts, p, k, lam_u = 30, PARAMS["Ts"], PARAMS["p"], PARAMS["k"], PARAMS["lambda_u"]
    A = np.array([[0, 1], [0, p]])
    Bm = np.array([0, k])
    w = np.ones(T) * Ts
    w[[0, -1]] *= .5
    wu = np.ones(T - 1) * Ts
    wu[[0, -1]] *= .5
    for seed in (0, 3):
        d = dcmotor_data(seed)
        _set_all(o, spec, d)
        st, it = o.solve(SOLVE_ARGS["mu0"], SOLVE_ARGS["maxIter"], -1, SOLVE_ARGS["addEye2Hessian"])
        y = o.get(c + "_getOutputs")
        assert st == 0
        names = [n for n, _ in spec.outputs]
        sizes = [int(np.prod(s)) if len(s) else 1 for _, s in spec.outputs]
        off = np.cumsum([0] + sizes)
        got = {n: y[off[i]:off[i + 1]] for i, n in enumerate(names)}
        u_t = got["u"].reshape((1, T - 1), order="F")
        x_t = got["x"].reshape((2, T), order="F")

        def unpack(z):
            return z[:T - 1], z[T - 1:].reshape((2, T), order="F")

        def cost(z):
            u, x = unpack(z)
            return (w * (x[0] - d["ref"]) ** 2).sum() + lam_u * (wu * u ** 2).sum()

        def eq(z):
            u, x = unpack(z)
            dyn = x[:, 1:] - (x[:, :-1] + Ts * (A @ x[:, :-1] + np.outer(Bm, u)))
            return np.r_[dyn.reshape(-1, order="F"), x[:, 0] - d["xinit"]]
        lo = np.r_[np.full(T - 1, -1.), np.tile([-np.inf, -np.inf], 1), np.tile([-.4, -.3], T - 1)]
        hi = np.r_[np.full(T - 1, 1.), np.tile([np.inf, np.inf], 1), np.tile([.4, .3], T - 1)]
        z0 = np.r_[u_t.reshape(-1), x_t.reshape(-1, order="F")]
        assert np.abs(eq(z0)).max() < 1e-4                        # equalTolerance
        assert (z0 >= lo - 1e-9).all() and (z0 <= hi + 1e-9).all()
        r = minimize(cost, z0, method="SLSQP", bounds=list(zip(lo, hi)),
                     constraints=[dict(type="eq", fun=eq)], options=dict(maxiter=500, ftol=1e-12))
        assert r.success
        assert abs(got["J"][0] - r.fun) <= 2e-4 * max(1, abs(r.fun))   # limited by desiredDualityGap/tolerances
        np.testing.assert_allclose(got["J"][0], cost(z0), rtol=1e-12)
        # warm-start outputs: shifted and clipped 5% inside the box (mpc_dcmotor.m:108-115)
        xw = got["xWarm"].reshape((2, T), order="F")
        assert (xw[0] >= -.4 + .05 * .8 - 1e-15).all() and (xw[1] <= .3 - .05 * .6 + 1e-15).all()
