"""GPU: the printed-code lowerings (one thread per instance; 8 lanes per instance with the bit-exact arithmetic) and the
two table-driven back-ends on tapes nobody arranged -- random sparse QPs with equalities and inequalities
(tenscalc_amd/problems/random_qp.py) -- against the oracle: status, iteration count and outputs bit for bit on every
instance, for three structures (one without equality constraints)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def libs():
    import __graft_entry__ as g
    return g.build()


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_qp_lowerings_bitwise(libs, tmp_path, seed):
    import __graft_entry__ as g
    from tenscalc_amd.problems.random_qp import SOLVE_ARGS, random_qp, random_qp_data
    from tenscalc_amd.runtime import HipSolver
    from tenscalc_amd.tapeio import save_tape
    from oracle.tc_oracle import Oracle
    spec = random_qp(seed, folder=g.GENERATED)
    tape = str(tmp_path / "p.tape")
    save_tape(spec.tape, tape)
    B = 203                                   # not a multiple of 8 or 64
    data = random_qp_data(seed, B)
    c = spec.classname
    names = [n for n, _ in spec.parameters + spec.variables]
    ref, rst, rit = Oracle(tape).solve_batch({f"{c}_set_{n}": data[n] for n in names}, c + "_getOutputs", B,
                                             mu0=SOLVE_ARGS["mu0"], maxIter=SOLVE_ARGS["maxIter"])
    assert (rst == 0).mean() > .9             # the problems are feasible and convex: the solver converges
    for v in ("strict", "w8x", "interp", "ldsm4", "ldsc4"):
        obj = HipSolver(libs["rqp%d_%s" % (seed, v)], B=B)
        for n in names:
            obj.set(n, data[n])
        st, it, _ = obj.solve(**SOLVE_ARGS)
        got = np.concatenate([o.reshape(B, -1) for o in obj.getOutputs()], axis=1)
        obj.close()
        assert np.array_equal(st, rst) and np.array_equal(it, rit), v
        assert np.array_equal(got, ref, equal_nan=True), v
