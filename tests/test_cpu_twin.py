"""CPU: the generated-C twin (tape printed as reference-style C + the reference's Newton loop) must be
bit-identical to the interpreter oracle -- two independent executions of the same tape."""
import numpy as np
import pytest


def _twin(spec):
    from oracle.c_emitter import CpuTwin
    try:
        return CpuTwin(spec)
    except (FileNotFoundError, OSError):
        pytest.skip("CPU twin needs /root/reference (or a prebuilt oracle/_ref library)")


def test_sls_twin_matches_interpreter(sls_spec, sls_tape_path):
    from tenscalc_amd.problems.sls import sls_batch
    from oracle.tc_oracle import Oracle
    tw = _twin(sls_spec)
    B = 300
    A, b, x0 = sls_batch(B, seed0=4000)
    At = np.transpose(A, (0, 2, 1)).reshape(B, -1)
    out, st, it = tw.solve_batch({"A": At, "b": b, "x": x0}, B, mu0=1.0, maxIter=30)
    c = sls_spec.classname
    ref, rst, rit = Oracle(sls_tape_path).solve_batch({c + "_set_A": At, c + "_set_b": b, c + "_set_x": x0},
                                                      c + "_getOutputs", B, mu0=1.0, maxIter=30)
    assert np.array_equal(out, ref) and np.array_equal(st, rst) and np.array_equal(it, rit)


def test_dcmotor_twin_matches_interpreter(native_built, tmp_path):
    import __graft_entry__ as g
    from tenscalc_amd.problems.mpc_dcmotor import SOLVE_ARGS, dcmotor_data, mpc_dcmotor
    from tenscalc_amd.tapeio import save_tape
    from oracle.tc_oracle import Oracle
    spec = mpc_dcmotor(folder=g.GENERATED)
    tw = _twin(spec)
    tape = str(tmp_path / "dc.tape")
    save_tape(spec.tape, tape)
    B = 20
    data = [dcmotor_data(50 + k) for k in range(B)]
    names = [n for n, _ in spec.parameters + spec.variables]
    stacked = {n: np.stack([np.asarray(d[n], dtype=float).reshape(-1, order="F") for d in data]) for n in names}
    out, st, it = tw.solve_batch(stacked, B, mu0=SOLVE_ARGS["mu0"], maxIter=SOLVE_ARGS["maxIter"])
    c = spec.classname
    ref, rst, rit = Oracle(tape).solve_batch({f"{c}_set_{n}": stacked[n] for n in names}, c + "_getOutputs", B,
                                             mu0=SOLVE_ARGS["mu0"], maxIter=SOLVE_ARGS["maxIter"])
    assert np.array_equal(out, ref) and np.array_equal(st, rst) and np.array_equal(it, rit)


@pytest.mark.parametrize("which", ["l1l2", "quadcopter"])
def test_big_tape_twins_reproduce_the_goldens(native_built, which):
    """the twins of the big tapes (every value in the scratchbook, group bodies in translation units compiled side by
    side) against the golden vectors, which came through the interpreter + the reference's Newton loop"""
    import os
    import __graft_entry__ as g
    from test_golden_configs import GOLD, _problem, _stack
    spec, data, args = _problem(which, g.GENERATED)
    tw = _twin(spec)
    gold = np.load(os.path.join(GOLD, which + ".npz"))
    names, st_ = _stack(spec, data, gold["seeds"])
    B = len(gold["seeds"])
    out, st, it = tw.solve_batch(st_, B, mu0=args["mu0"], maxIter=args["maxIter"])
    assert np.array_equal(st, gold["status"]) and np.array_equal(it, gold["iter"]) and np.array_equal(out, gold["outputs"])
