"""CPU, development container only: the restated Newton loop (oracle/tc_oracle.c tco_solve) against
the reference's own lib/ipmPD_CSsolver.c compiled in place and linked to the same tape interpreter
(oracle/_ref).  Skipped where neither /root/reference nor a prebuilt oracle/_ref exists."""
import numpy as np
import pytest

from tenscalc_amd.problems.sls import sls_data


@pytest.fixture(scope="module")
def both(sls_spec, sls_tape_path):
    from oracle.tc_oracle import Oracle, ReferenceLoopOracle
    try:
        ro = ReferenceLoopOracle(sls_tape_path, "sls", sls_spec.defines)
    except (FileNotFoundError, OSError):
        pytest.skip("oracle/_ref not available (no /root/reference and no prebuilt library)")
    return Oracle(sls_tape_path), ro, sls_spec.classname


@pytest.mark.parametrize("mu0,maxIter,eps", [(1.0, 30, 1e-9), (1e-3, 30, 1e-9), (1.0, 5, 1e-9),
                                             (10.0, 60, 1e-3), (1.0, 30, 1e-20),
                                             (1.0, 30, -10.0), (1.0, 40, -1.0)])   # < 0: negative curvature, the
#                                              second pass of the regularisation loop (ipmPD_CSsolver.c:459-531) runs
def test_status_iter_outputs_bitwise(both, mu0, maxIter, eps):
    o, ro, c = both
    for seed in range(100, 140):
        A, b, x0 = sls_data(seed)
        r = []
        for q in (o, ro):
            q.set(c + "_set_A", A); q.set(c + "_set_b", b); q.set(c + "_set_x", x0)
            st, it = q.solve(mu0, maxIter, -1, (eps, eps))
            r.append((st, it, q.get(c + "_getOutputs")))
        assert r[0][:2] == r[1][:2], (seed, r[0][:2], r[1][:2])
        assert np.array_equal(r[0][2], r[1][2], equal_nan=True)


def test_degenerate_inputs(both):
    o, ro, c = both
    A, b, x0 = sls_data(7)
    cases = [(A * 0, b, x0), (A, b * 1e6, x0), (A * 1e-8, b, x0), (A, b, np.full(8, .05 - 1e-13)),
             (A, b, np.full(8, 1e-300)), (A * np.nan, b, x0), (A, b, x0 * 0)]
    for (a_, b_, x_) in cases:
        r = []
        for q in (o, ro):
            q.set(c + "_set_A", a_); q.set(c + "_set_b", b_); q.set(c + "_set_x", x_)
            st, it = q.solve(1.0, 30, -1, (1e-9, 1e-9))
            r.append((st, it, q.get(c + "_getOutputs")))
        assert r[0][:2] == r[1][:2]
        assert np.array_equal(r[0][2], r[1][2], equal_nan=True)
