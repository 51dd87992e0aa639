"""The reference's solverVerboseLevel >= 3 iteration table (lib/ipmPD_CSsolver.c:248-345,384,422-546,775-863), recorded for
one instance of a batch by the device loop (csrc/ipm_newton.hip.inc: TC_V3) and written by tc_trace.  The numbers are
checked against tests/golden/sls_trace.npz, which was produced by the reference's own ipmPD_CSsolver.c stopped at
maxIter = k: row k+1 of the table prints the state after k Newton iterations."""
import ctypes
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _golden(seed):
    g = np.load(os.path.join(GOLD, "sls_trace.npz"))
    rows = g["rows"]
    return rows[rows[:, 0] == seed], int(g["iters"][list(np.unique(rows[:, 0])).index(seed)])


def _check_rows(rows, seed):
    """rows [n, 24] recorded by the loop against the golden trace of `seed` (columns: ipm_newton.hip.inc)"""
    gold, iters = _golden(seed)
    assert rows.shape[0] == iters and np.array_equal(rows[:, 0], np.arange(1, iters + 1))
    assert rows[-1, 7] == 1 and (rows[:-1, 7] == 0).all()                 # clean exit in the last row
    for k in range(1, iters):                                             # state after k iterations = head of pass k+1
        gk, r = gold[k - 1], rows[k]
        assert int(gk[1]) == k
        assert r[1] == gk[4] and r[2] == gk[13]                           # J, |grad|
        assert r[6] == gk[14] and r[4] == gk[15] and r[5] == gk[16]       # gap, min F, min lambda
        assert rows[k - 1, 17] == gk[18]                                  # alphaPrimal of pass k
        if k < iters - 1:
            assert r[8] == gk[17]                                         # mu printed by pass k+1 = mu after k updates
    assert rows[0, 8] == 1.0 and (np.diff(rows[:-1, 8]) <= 0).all()       # mu0, then decreasing
    assert set(rows[:-1, 20]) <= {1.0, 2.0, 3.0, 4.0} and (rows[:-1, 12] >= 0).all()


@pytest.mark.parametrize("seed", [0, 5])
def test_device_loop_on_the_host_records_the_reference_table(native_built, tmp_path, seed):
    from tenscalc_amd import tcalculus as tc
    from tenscalc_amd.problems.sls import SOLVE_ARGS, sls, sls_data
    from tenscalc_amd.tapeio import save_tape
    from oracle.tc_oracle import DeviceLoopOracle
    tc.clear()
    spec = sls(classname="tmp_sls_v3", solverVerboseLevel=3)
    assert spec.defines["verboseLevel"] == 3
    tape = str(tmp_path / "v3.tape")
    save_tape(spec.tape, tape)
    o = DeviceLoopOracle(tape, "sls_v3", spec.defines)
    A, b, x0 = sls_data(seed)
    c = spec.classname
    o.set(c + "_set_A", A); o.set(c + "_set_b", b); o.set(c + "_set_x", x0)
    st, it = o.solve(**SOLVE_ARGS)
    buf = np.zeros((4096, 24))
    o.L.tcdev_v3_rows.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.c_int]
    n = o.L.tcdev_v3_rows(buf.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), 4096)
    assert st == 0 and n == it
    _check_rows(buf[:n], seed)
    # recording changes nothing: the plain loop returns the same status / iteration count / outputs
    tc.clear()
    plain = sls(classname="tmp_sls_v0")
    tape0 = str(tmp_path / "v0.tape")
    save_tape(plain.tape, tape0)
    o0 = DeviceLoopOracle(tape0, "sls_v0", plain.defines)
    c0 = plain.classname
    o0.set(c0 + "_set_A", A); o0.set(c0 + "_set_b", b); o0.set(c0 + "_set_x", x0)
    assert o0.solve(**SOLVE_ARGS) == (st, it)
    assert np.array_equal(o0.get(c0 + "_getOutputs"), o.get(c + "_getOutputs"))


@pytest.mark.gpu
def test_tc_trace_writes_the_table_for_one_instance_of_a_batch(tmp_path):
    import __graft_entry__ as g
    from tenscalc_amd import tcalculus as tc
    from tenscalc_amd.problems.sls import SOLVE_ARGS, sls, sls_batch
    from tenscalc_amd.runtime import HipSolver, TensCalcHipError
    tc.clear()
    spec = sls(classname="tmp_sls_v3", solverVerboseLevel=3, folder=g.GENERATED)
    B = 96
    A, b, x0 = sls_batch(B)                       # instance k = seed k: 0 and 5 are golden seeds
    obj = HipSolver(spec.build(**dict(g.VARIANT_KW["strict"], tag="strict_v3")), B=B)
    with pytest.raises(TensCalcHipError, match="call tc_solve first"):
        obj.trace(0, str(tmp_path / "none.txt"))
    obj.set("A", np.ascontiguousarray(np.transpose(A, (0, 2, 1)))); obj.set("b", b); obj.set("x", x0)
    st, it, _ = obj.solve(**SOLVE_ARGS)
    outs = [o_.copy() for o_ in obj.getOutputs()]
    for seed in (0, 5):
        path = str(tmp_path / ("trace%d.txt" % seed))
        obj.trace(seed, path)
        txt = open(path).read().splitlines()
        gold, iters = _golden(seed)
        assert txt[2].startswith("Iter      cost   |grad|   |eq|    ineq.    dual    gap   l(mu) l(Hp) l(Hd) curvature  d.err. ")
        assert txt[3].startswith("  30:<-mx tol.-> 1.0e-04 1.0e-04")
        lines = txt[4:4 + iters]
        assert lines[-1].endswith("us") and "-> clean exit" in lines[-1] and it[seed] == iters
        for k in range(1, iters):                 # "%4d:%11.3e%8.1e  -eq-  %8.1e%8.1e%8.1e%6.1f..."
            ln, gk = lines[k], gold[k - 1]
            assert ln[:5] == "%4d:" % (k + 1)
            assert ln[5:16] == "%11.3e" % gk[4] and ln[16:24] == "%8.1e" % gk[13] and ln[24:32] == "  -eq-  "
            assert ln[32:56] == "%8.1e%8.1e%8.1e" % (gk[15], gk[16], gk[14])
            if k < iters - 1:
                assert ln[56:62] == "%6.1f" % np.log10(gk[17])
        assert txt[4 + iters] == "%4d:status=0x0" % iters
    assert np.array_equal(obj.getOutputs()[1], outs[1])          # tracing solved the batch again: same results
    with pytest.raises(TensCalcHipError, match="outside the batch"):
        obj.trace(B, str(tmp_path / "x.txt"))
    obj.close()
    plain = HipSolver(g.sls_spec("strict").build(**g.VARIANT_KW["strict"]), B=8)
    with pytest.raises(TensCalcHipError, match="solverVerboseLevel < 3"):
        plain.trace(0, str(tmp_path / "y.txt"))
    plain.close()
