"""the reference's ORIGINAL assembly lib/ipmPD_CS.m -- additive lambda update, optional Mehrotra
predictor ("affine") step (skipAffine=false, ipmPD_CSsolver.c:576-665) and the smaller Newton matrix -- through
tenscalc_amd (cmex2optimizeCS(..., lambdaUpdate="additive")).

CPU: (1) the oracle's restated loop against the REFERENCE's own ipmPD_CSsolver.c compiled in place with the matching -D
block (-DskipAffine=0, -DsmallerNewtonMatrix=1): status, iteration count and outputs bit for bit; (2) the DEVICE loop
(csrc/ipm_newton.hip.inc) compiled for the host, likewise; (3) the optimum against scipy.
GPU: the strict and w8x lowerings bit for bit against the oracle."""
import numpy as np
import pytest

from tenscalc_amd.problems.sls import SOLVE_ARGS, sls_data

VARIANTS = {"add": dict(), "aff": dict(skipAffine=False), "small": dict(smallerNewtonMatrix=True),
            "smallaff": dict(skipAffine=False, smallerNewtonMatrix=True)}


def _spec(v, folder):
    from tenscalc_amd.problems.sls import sls
    return sls(classname="tmp_sls_" + v, folder=folder, lambdaUpdate="additive", **VARIANTS[v])


def _eq_problem(v, folder):
    """equality + inequality constraints (nG > 0): the norminf_eq condition of the sigma update, setAlphaDualEq__"""
    from tenscalc_amd import tcalculus as tc
    from tenscalc_amd.cmex2optimizeCS import cmex2optimizeCS
    A = tc.Tvariable("A", (12, 4)); b = tc.Tvariable("b", (12,)); x = tc.Tvariable("x", (4,))
    return cmex2optimizeCS(classname="tmp_eq_" + v, objective=tc.norm2(A * x - b) / 12, optimizationVariables=[x],
                           constraints=[tc.tsum(x) == .1, x >= 0], outputExpressions=[tc.norm2(A * x - b) / 12, x],
                           outputNames=["J", "x"], parameters=[A, b], folder=folder, lambdaUpdate="additive", **VARIANTS[v])


@pytest.mark.parametrize("v", list(VARIANTS))
def test_additive_variants_oracle_vs_reference_loop(native_built, tmp_path, v):
    from tenscalc_amd.tapeio import save_tape
    from oracle.tc_oracle import DeviceLoopOracle, Oracle, ReferenceLoopOracle
    from scipy.optimize import lsq_linear
    spec = _spec(v, str(tmp_path))
    assert spec.defines["skipAffine"] == int(not VARIANTS[v].get("skipAffine") is False)
    tape = str(tmp_path / "t.tape")
    save_tape(spec.tape, tape)
    try:
        ro = ReferenceLoopOracle(tape, "sls_" + v, spec.defines)
    except (FileNotFoundError, OSError):
        pytest.skip("oracle/_ref not available (no /root/reference and no prebuilt library)")
    qs = [Oracle(tape), ro, DeviceLoopOracle(tape, "sls_" + v, spec.defines)]
    c = spec.classname
    its = []
    for mu0, maxIter, eps in [(1.0, 30, 1e-9), (1e-3, 30, 1e-9), (1.0, 4, 1e-9), (1.0, 40, -1.0)]:
        for seed in range(200, 225):
            A, b, x0 = sls_data(seed)
            r = []
            for q in qs:
                q.set(c + "_set_A", A); q.set(c + "_set_b", b); q.set(c + "_set_x", x0)
                st, it = q.solve(mu0, maxIter, -1, (eps, eps))
                r.append((st, it, q.get(c + "_getOutputs")))
            for k in (1, 2):
                assert r[0][:2] == r[k][:2], (v, seed, k, r[0][:2], r[k][:2])
                assert np.array_equal(r[0][2], r[k][2], equal_nan=True), (v, seed, k)
            if (mu0, maxIter, eps) == (1.0, 30, 1e-9):
                assert r[0][0] == 0
                its.append(r[0][1])
                ref = lsq_linear(A, b, bounds=(0, .05), tol=1e-12)
                assert np.abs(r[0][2][1:] - ref.x).max() <= 5e-4          # gradTolerance 1e-4, duality gap 1e-5
    print("\n%s: iterations %d..%d" % (v, min(its), max(its)))


@pytest.mark.parametrize("v", ["aff", "smallaff"])
def test_affine_step_with_equalities(native_built, tmp_path, v):
    from tenscalc_amd.tapeio import save_tape
    from oracle.tc_oracle import DeviceLoopOracle, Oracle, ReferenceLoopOracle
    spec = _eq_problem(v, str(tmp_path))
    assert spec.defines["nG"] == 1 and spec.defines["skipAffine"] == 0
    tape = str(tmp_path / "t.tape")
    save_tape(spec.tape, tape)
    try:
        ro = ReferenceLoopOracle(tape, "eq_" + v, spec.defines)
    except (FileNotFoundError, OSError):
        pytest.skip("oracle/_ref not available")
    qs = [Oracle(tape), ro, DeviceLoopOracle(tape, "eq_" + v, spec.defines)]
    c = spec.classname
    rs = np.random.RandomState(3)
    for k in range(30):
        A = rs.random_sample((12, 4)); b = rs.random_sample(12); x0 = .02 * rs.random_sample(4) + .005
        r = []
        for q in qs:
            q.set(c + "_set_A", A); q.set(c + "_set_b", b); q.set(c + "_set_x", x0)
            st, it = q.solve(1e-2, 40, -1, (1e-9, 1e-9))
            r.append((st, it, q.get(c + "_getOutputs")))
        assert r[0][:2] == r[1][:2] == r[2][:2] and r[0][0] == 0
        assert np.array_equal(r[0][2], r[1][2]) and np.array_equal(r[0][2], r[2][2])
        assert abs(r[0][2][1:].sum() - .1) <= 1e-4


def test_multiplicative_variant_still_rejects_them():
    from tenscalc_amd.problems.sls import sls
    with pytest.raises(NotImplementedError):
        sls(folder="/tmp/x", skipAffine=False)
    with pytest.raises(NotImplementedError):
        sls(folder="/tmp/x", smallerNewtonMatrix=True)


@pytest.mark.gpu
@pytest.mark.parametrize("v", ["aff", "smallaff"])
@pytest.mark.parametrize("lowering", ["strict", "w8x"])
def test_additive_variants_on_gpu(tmp_path, v, lowering):
    import __graft_entry__ as g
    from tenscalc_amd.problems.sls import sls_batch
    from tenscalc_amd.runtime import HipSolver
    from tenscalc_amd.tapeio import save_tape
    from oracle.tc_oracle import Oracle
    spec = _spec(v, g.GENERATED)
    so = spec.build(**g.VARIANT_KW[lowering])
    B = 1000
    A, b, x0 = sls_batch(B, seed0=31000)
    A[3] *= np.nan
    x0[5] = .06
    tape = str(tmp_path / "t.tape")
    save_tape(spec.tape, tape)
    c = spec.classname
    ref, rst, rit = Oracle(tape).solve_batch({c + "_set_A": np.transpose(A, (0, 2, 1)).reshape(B, -1), c + "_set_b": b,
                                              c + "_set_x": x0}, c + "_getOutputs", B, mu0=SOLVE_ARGS["mu0"],
                                             maxIter=SOLVE_ARGS["maxIter"])
    obj = HipSolver(so, B=B, device=0)
    obj.setP_A(np.ascontiguousarray(np.transpose(A, (0, 2, 1)))); obj.setP_b(b); obj.setV_x(x0)
    st, it, _ = obj.solve(**SOLVE_ARGS)
    J, x = obj.getOutputs()
    assert np.array_equal(st, rst) and np.array_equal(it, rit)
    assert np.array_equal(J.reshape(B), ref[:, 0], equal_nan=True) and np.array_equal(x.reshape(B, 8), ref[:, 1:], equal_nan=True)
    assert (st == 0).sum() >= B - 2


def _sens_data(seed, N=12, n=3, ns=2):
    rng = np.random.RandomState(seed)
    return rng.rand(N, ns + n), rng.rand(N), .1 * rng.rand(n), .1 * rng.rand(ns)


def test_sensitivity_outputs(native_built, tmp_path):
    """cmex2optimizeCS(..., sensitivityVariables=[p]) (lib/ipmPD_CS.m:478-503, variableIndices.m): the outputs Du1,
    DfDu1, D2fDu1 are what the formulas say, evaluated on the solver's own Hess and Lf_u at the returned point --
    Hess1*Du1 = B1 with Hess1 / B1 the rows and columns of WW that the flags select, D2fDu1 = WW11 + WW1z*Du1.
    Tape evaluation against dense numpy algebra (the reference has no fixture for this option)."""
    from tenscalc_amd.problems.small import sensitivity_ls
    from tenscalc_amd.tapeio import save_tape
    from oracle.tc_oracle import Oracle
    from tenscalc_amd import tcalculus as tc
    from tenscalc_amd.cmex2optimizeCS import cmex2optimizeCS
    spec = sensitivity_ls(folder=str(tmp_path))
    assert spec.defines["nU"] == 5 and spec.defines["nG"] == 1 and spec.defines["nF"] == 3
    tape = str(tmp_path / "t.tape")
    save_tape(spec.tape, tape)
    o = Oracle(tape)
    c = spec.classname
    sizes = [int(np.prod(s)) if len(s) else 1 for _, s in spec.outputs]
    for seed in range(6):
        A, b, x0, p0 = _sens_data(seed)
        o.set(c + "_set_A", A); o.set(c + "_set_b", b); o.set(c + "_set_x", x0); o.set(c + "_set_p", p0)
        st, it = o.solve(1.0, 40, -1, (1e-9, 1e-9))
        assert st == 0
        y = np.split(o.get(c + "_getOutputs"), np.cumsum(sizes)[:-1])
        H = y[3].reshape(9, 9, order="F")
        Lf_u, Du1, DfDu1, D2 = y[4], y[5].reshape(7, 2, order="F"), y[6], y[7].reshape(2, 2, order="F")
        s1, rest = [0, 1], list(range(2, 9))
        np.testing.assert_array_equal(DfDu1, Lf_u[s1])
        ref = np.linalg.solve(H[np.ix_(rest, rest)], -H[np.ix_(rest, s1)])
        np.testing.assert_allclose(Du1, ref, rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(D2, H[np.ix_(s1, s1)] + H[np.ix_(s1, rest)] @ ref, rtol=1e-8, atol=1e-10)
    # the flags: a subsref of an optimization variable selects entries; errors as in the reference
    A_ = tc.Tvariable("A", (6, 4)); b_ = tc.Tvariable("b", (6,)); x_ = tc.Tvariable("x", (4,))
    kw = dict(objective=tc.norm2(A_ * x_ - b_), optimizationVariables=[x_], constraints=[x_ >= 0], parameters=[A_, b_],
              outputExpressions=[x_, tc.Tvariable("D2fDu1_", (2, 2))], outputNames=["x", "D2"], folder=str(tmp_path))
    sp = cmex2optimizeCS(classname="tmp_sens2", sensitivityVariables=[x_[1:3]], lambdaUpdate="additive", **kw)
    assert sp.outputs[1][1] == (2, 2)
    with pytest.raises(NotImplementedError, match="ipmPD_CS.m"):
        cmex2optimizeCS(classname="tmp_sens3", sensitivityVariables=[x_], **kw)
    with pytest.raises(NotImplementedError, match="smallerNewtonMatrix"):
        cmex2optimizeCS(classname="tmp_sens4", sensitivityVariables=[x_[0:2]], lambdaUpdate="additive", smallerNewtonMatrix=True, **kw)
    with pytest.raises(ValueError, match="not an optimizationVariable"):
        cmex2optimizeCS(classname="tmp_sens5", sensitivityVariables=[b_], lambdaUpdate="additive", **kw)


@pytest.mark.gpu
@pytest.mark.parametrize("lowering", ["strict", "interp"])
def test_sensitivity_outputs_on_gpu(tmp_path, lowering):
    """the sensitivity outputs (a second LDL with a matrix right-hand side, evaluated in getOutputs) bit for bit
    against the oracle"""
    import __graft_entry__ as g
    from tenscalc_amd.problems.small import sensitivity_ls
    from tenscalc_amd.runtime import HipSolver
    from tenscalc_amd.tapeio import save_tape
    from oracle.tc_oracle import Oracle
    spec = sensitivity_ls(folder=g.GENERATED)
    so = spec.build(**g.VARIANT_KW[lowering])
    B = 300
    data = [_sens_data(500 + k) for k in range(B)]
    A = np.stack([d[0] for d in data]); b = np.stack([d[1] for d in data])
    x0 = np.stack([d[2] for d in data]); p0 = np.stack([d[3] for d in data])
    tape = str(tmp_path / "t.tape")
    save_tape(spec.tape, tape)
    c = spec.classname
    ref, rst, rit = Oracle(tape).solve_batch({c + "_set_A": np.transpose(A, (0, 2, 1)).reshape(B, -1), c + "_set_b": b,
                                              c + "_set_x": x0, c + "_set_p": p0}, c + "_getOutputs", B, mu0=1.0, maxIter=40)
    obj = HipSolver(so, B=B, device=0)
    obj.setP_A(np.ascontiguousarray(np.transpose(A, (0, 2, 1)))); obj.setP_b(b); obj.setV_x(x0); obj.setV_p(p0)
    st, it, _ = obj.solve(mu0=1.0, maxIter=40, saveIter=-1, addEye2Hessian=(1e-9, 1e-9))
    got = np.concatenate([np.asarray(y).reshape(B, -1) for y in obj.getOutputs()], axis=1)
    assert np.array_equal(st, rst) and np.array_equal(it, rit) and (st == 0).all()
    assert np.array_equal(got, ref, equal_nan=True)
