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
