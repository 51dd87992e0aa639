"""CPU: the DEVICE Newton loop (tenscalc_amd/csrc/ipm_newton.hip.inc, the file every generated
HIP class includes) compiled for the host over the oracle's tape interpreter
(oracle/devloop_glue.cpp) against the oracle's restatement of lib/ipmPD_CSsolver.c -- which
tests/test_oracle_vs_reference_loop.py and tests/test_solver_options.py pin bit-for-bit to the
reference file itself.  The device loop folds the reference's repeated call sites into loops (one
call site per entry point); this is the test that the folding changed nothing: status, iteration
count and outputs identical on sls (incl. degenerate inputs, maxIter exits, NaN) and on every
compile-time branch of the template."""
import numpy as np
import pytest

from tenscalc_amd.problems.sls import sls_data
from test_solver_options import OPTIONS, _problem


@pytest.fixture(scope="module")
def both(sls_spec, sls_tape_path):
    from oracle.tc_oracle import DeviceLoopOracle, Oracle
    return Oracle(sls_tape_path), DeviceLoopOracle(sls_tape_path, "sls", sls_spec.defines), sls_spec.classname


@pytest.mark.parametrize("mu0,maxIter,eps", [(1.0, 30, 1e-9), (1e-3, 30, 1e-9), (1.0, 5, 1e-9), (1.0, 1, 1e-9),
                                             (10.0, 60, 1e-3), (1.0, 30, 1e-20), (1e3, 12, 1e-9),
                                             (1.0, 30, -10.0), (1.0, 40, -1.0)])   # < 0: second regularisation pass
def test_sls_status_iter_outputs_bitwise(both, mu0, maxIter, eps):
    o, do, c = both
    stats = set()
    for seed in range(100, 160):
        A, b, x0 = sls_data(seed)
        r = []
        for q in (o, do):
            q.set(c + "_set_A", A); q.set(c + "_set_b", b); q.set(c + "_set_x", x0)
            st, it = q.solve(mu0, maxIter, -1, (eps, eps))
            r.append((st, it, q.get(c + "_getOutputs")))
        assert r[0][:2] == r[1][:2], (seed, r[0][:2], r[1][:2])
        assert np.array_equal(r[0][2], r[1][2], equal_nan=True)
        stats.add(r[0][0])
    if maxIter <= 5:
        assert any(s & 8 for s in stats)          # the maxIter epilogue was exercised


def test_sls_degenerate_inputs(both):
    o, do, c = both
    A, b, x0 = sls_data(7)
    cases = [(A * 0, b, x0), (A, b * 1e6, x0), (A * 1e-8, b, x0), (A, b, np.full(8, .05 - 1e-13)),
             (A, b, np.full(8, 1e-300)), (A * np.nan, b, x0), (A, b, x0 * 0), (A, b, np.full(8, .06)),
             (A * 1e8, b, x0), (A, b * np.nan, x0)]
    seen = set()
    for (a_, b_, x_) in cases:
        r = []
        for q in (o, do):
            q.set(c + "_set_A", a_); q.set(c + "_set_b", b_); q.set(c + "_set_x", x_)
            st, it = q.solve(1.0, 30, -1, (1e-9, 1e-9))
            r.append((st, it, q.get(c + "_getOutputs")))
        assert r[0][:2] == r[1][:2]
        assert np.array_equal(r[0][2], r[1][2], equal_nan=True)
        seen.add(r[0][0])
    assert len(seen) >= 3                          # converged, NaN and infeasible-start exits all hit


@pytest.mark.parametrize("kind", ["box", "eq", "eqonly", "free"])
@pytest.mark.parametrize("opt", OPTIONS, ids=[str(k) for k in range(len(OPTIONS))])
def test_template_branches_bitwise(native_built, tmp_path, kind, opt):
    from tenscalc_amd.tapeio import save_tape
    from oracle.tc_oracle import DeviceLoopOracle, Oracle
    if kind in ("eqonly", "free") and ("scaleInequalities" in opt or "coupledAlphas" in opt):
        pytest.skip("option only matters with inequality constraints")
    spec = _problem(kind, **opt)
    tape = str(tmp_path / "p.tape")
    save_tape(spec.tape, tape)
    do = DeviceLoopOracle(tape, "opt_%s_%d" % (kind, OPTIONS.index(opt)), spec.defines)
    o = Oracle(tape)
    c = spec.classname
    rs = np.random.RandomState(3)
    for trial in range(8):
        A, b, x0 = rs.random_sample((12, 4)), rs.random_sample(12), .02 * rs.random_sample(4) + .005
        if trial >= 6:
            A = A * 1e4                           # badly scaled: regularisation retries, small steps
        r = []
        ae = (-5.0, -5.0) if trial == 4 else (1e-9, 1e-9)     # negative: second regularisation pass
        for q in (o, do):
            q.set(c + "_set_A", A); q.set(c + "_set_b", b); q.set(c + "_set_x", x0)
            st, it = q.solve(1.0 if trial % 2 else 1e-2, 40 if trial != 5 else 3, -1, ae)
            r.append((st, it, q.get(c + "_getOutputs")))
        assert r[0][:2] == r[1][:2], (kind, opt, trial, r[0][:2], r[1][:2])
        assert np.array_equal(r[0][2], r[1][2], equal_nan=True)
