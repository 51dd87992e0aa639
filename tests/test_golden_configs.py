"""The committed golden vectors of BASELINE configs[2..4] and the per-iteration sls trace (tests/golden/*.npz,
generated through the REFERENCE's own Newton loop by tests/golden/make_golden_configs.py).

What these fixtures are -- and are not.  They are outputs of THIS repo's tape, walked by the oracle's interpreter under the
reference's Newton loop: they move whenever the tape does (the elimination order of a class changed in rounds 3 and 5 and they
were regenerated), so they catch drift between the HIP back-ends and the oracle, NOT a regression of the front-end that builds
the tape.  What guards the front-end is independent of any tape: the dense interior-point twins of tests/dense_ipm.py
(tests/test_tape_vs_dense_ipm.py, test_random_qp_vs_dense_ipm.py: same iterates to 1e-7 .. 1e-10, with the number of seeds
allowed to leave the dense path written in the test), the reference's flop table (tests/test_reference_flops_table.py) and the
reference's own emitter printing this repo's tapes (tests/test_reference_emitter_replay.py).

CPU part: the oracle (restated loop) reproduces every fixture bit for bit -- the fixtures and the checker agree.
GPU part: the HIP back-ends, called through the C-ABI, reproduce them -- bit for bit for mpc_dcmotor, l1l2 and the
sls trace, and since round 4 for mpc_quadcopter too (its pow calls are evaluated as the host libm does, csrc/tc_glibcpow.hip.inc)."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _problem(which, folder):
    if which == "dcmotor":
        from tenscalc_amd.problems.mpc_dcmotor import SOLVE_ARGS, dcmotor_data as data, mpc_dcmotor as mk
    elif which == "l1l2":
        from tenscalc_amd.problems.l1l2 import SOLVE_ARGS, l1l2 as mk, l1l2_data as data
    else:
        from tenscalc_amd.problems.mpc_quadcopter import SOLVE_ARGS, mpc_quadcopter as mk, quadcopter_data as data
    return mk(folder=folder), data, SOLVE_ARGS


def _stack(spec, data, seeds):
    names = [n for n, _ in spec.parameters + spec.variables]
    ds = [data(int(s)) for s in seeds]
    return names, {n: np.stack([np.asarray(d[n], dtype=float).reshape(-1, order="F") for d in ds]) for n in names}


@pytest.mark.parametrize("which", ["dcmotor", "l1l2", "quadcopter"])
def test_oracle_reproduces_golden(native_built, tape_oracle, tmp_path, which):
    g = np.load(os.path.join(GOLD, which + ".npz"))
    spec, data, args = _problem(which, str(tmp_path))
    assert list(g["output_names"]) == [nm for nm, _ in spec.outputs]
    names, st_ = _stack(spec, data, g["seeds"])
    chk = sum((k + 1) * st_[n].sum(axis=1) for k, n in enumerate(names))
    assert np.allclose(chk, g["checksum"], rtol=1e-13)                     # the inputs are the ones the fixture was made from
    c = spec.classname
    o = tape_oracle(spec.tape, c)
    B = len(g["seeds"])
    ref, rst, rit = o.solve_batch({f"{c}_set_{n}": st_[n] for n in names}, c + "_getOutputs", B,
                                  mu0=args["mu0"], maxIter=args["maxIter"])
    assert np.array_equal(rst, g["status"]) and np.array_equal(rit, g["iter"])
    assert np.array_equal(ref, g["outputs"], equal_nan=True)
    if which == "quadcopter":
        ref1, _, _ = o.solve_batch({f"{c}_set_{n}": st_[n] for n in names}, c + "_getOutputs", B, mu0=args["mu0"], maxIter=1)
        assert np.array_equal(ref1, g["outputs_maxIter1"])


def test_oracle_reproduces_sls_trace(sls_spec, tape_oracle):
    from tenscalc_amd.problems.sls import SOLVE_ARGS, sls_data
    g = np.load(os.path.join(GOLD, "sls_trace.npz"))
    rows = g["rows"]
    c = sls_spec.classname
    o = tape_oracle(sls_spec.tape, c)
    for r in rows[::7]:
        A, b, x0 = sls_data(int(r[0]))
        o.set(c + "_set_A", A); o.set(c + "_set_b", b); o.set(c + "_set_x", x0)
        st, it = o.solve(SOLVE_ARGS["mu0"], int(r[1]), -1, SOLVE_ARGS["addEye2Hessian"])
        assert (st, it) == (int(r[2]), int(r[3]))
        assert np.array_equal(o.get(c + "_getOutputs"), r[4:13])
        assert o.get("getNorminf_Grad__")[0] == r[13] and np.array_equal(o.get("getGapMinFMinLambda__"), r[14:17])
    # the trace is a real trace: mu decreases along a solve, the last row of a seed is close to convergence
    k0 = int(g["first_row"][0])
    n0 = int(g["iters"][0]) - 1
    mu = rows[k0:k0 + n0, 17]
    assert mu[0] <= 1.0 and mu[-1] < 1e-3 and (np.diff(mu) <= 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("which,tags", [("dcmotor", ["dcmotor_strict", "dcmotor_w32x", "dcmotor_interp", "dcmotor_ldsm4", "dcmotor_ldsc", "dcmotor_ldsc4"]),
                                        ("l1l2", ["l1l2_interp", "l1l2_ldsm4", "l1l2_ldsc4"]),
                                        ("quadcopter", ["quadcopter_ldsm4", "quadcopter_ldsm8", "quadcopter_ldsc4", "quadcopter_ldsc4w", "quadcopter_interp"])])
def test_gpu_reproduces_golden(which, tags):
    import __graft_entry__ as g_
    from tenscalc_amd.runtime import HipSolver
    libs = g_.build()
    g = np.load(os.path.join(GOLD, which + ".npz"))
    spec, data, args = _problem(which, g_.GENERATED)
    names, st_ = _stack(spec, data, g["seeds"])
    B = len(g["seeds"])
    for tag in tags:
        obj = HipSolver(libs[tag], B=B, device=0)
        for n in names:
            obj.set(n, st_[n])
        if which == "quadcopter":
            # its tape calls pow(x,-.5), pow(x,-1.5); the device evaluates them as the host libm does, bit for bit
            # (csrc/tc_glibcpow.hip.inc), so the complete solves -- 20..371 Newton iterations on a non-convex
            # problem -- reproduce the goldens like every other config.  First the outputs after ONE iteration.
            obj.solve(**dict(args, maxIter=1))
            got = np.concatenate([a.reshape(B, -1) for a in obj.getOutputs()], axis=1)
            assert np.array_equal(got, g["outputs_maxIter1"]), tag
        st, it, _ = obj.solve(**args)
        got = np.concatenate([a.reshape(B, -1) for a in obj.getOutputs()], axis=1)
        assert np.array_equal(st, g["status"]) and np.array_equal(it, g["iter"]), (tag, it, g["iter"])
        assert np.array_equal(got, g["outputs"]), tag
        obj.close()


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["strict", "w8x"])
def test_gpu_reproduces_sls_trace(variant):
    """every row of the per-iteration trace: the solve stopped by maxIter=k returns the reference's status, iteration
    count, J and x bit for bit (the batch holds one instance per trace row, solved with that row's maxIter)"""
    import __graft_entry__ as g_
    from tenscalc_amd.problems.sls import SOLVE_ARGS, sls_data
    from tenscalc_amd.runtime import HipSolver
    spec = g_.sls_spec(variant)
    so = spec.build(**g_.VARIANT_KW[variant])
    g = np.load(os.path.join(GOLD, "sls_trace.npz"))
    rows = g["rows"]
    for mi in sorted(set(rows[:, 1].astype(int))):
        sel = rows[rows[:, 1] == mi]
        B = len(sel)
        A = np.empty((B, 8, 100)); b = np.empty((B, 100)); x0 = np.empty((B, 8))
        for q, r in enumerate(sel):
            a_, b_, x_ = sls_data(int(r[0]))
            A[q], b[q], x0[q] = a_.T, b_, x_
        obj = HipSolver(so, B=B, device=0)
        obj.setP_A(A); obj.setP_b(b); obj.setV_x(x0)
        st, it, _ = obj.solve(**dict(SOLVE_ARGS, maxIter=int(mi)))
        J, x = obj.getOutputs()
        assert np.array_equal(st, sel[:, 2].astype(np.int32)) and np.array_equal(it, sel[:, 3].astype(np.int32)), mi
        assert np.array_equal(J.reshape(B), sel[:, 4]) and np.array_equal(x.reshape(B, 8), sel[:, 5:13]), mi
        obj.close()
