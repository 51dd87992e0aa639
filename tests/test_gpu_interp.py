"""GPU parity of the table-driven back-end (body='interp': the tape as data, csrc/tc_interp.hip.inc),
called through the C-ABI, against the CPU oracle.

Tolerances.  The walk evaluates every instruction in the reference's operand order with IEEE fp64
+,-,*,/,sqrt (built with -ffp-contract=off), so for tapes made of those operations -- sls,
mpc_dcmotor, l1l2 (BASELINE configs 1-4) -- outputs, status and iteration counts must be BIT-IDENTICAL
to the oracle.  Tapes that call libm transcendentals (mpc_quadcopter: .5*pow(x,-.5), -.25*pow(x,-1.5))
cannot be: the device pow() and the host's differ by 1 ulp on about a quarter of their arguments
(scripts/powcheck), and on this non-convex problem one ulp can send the iteration down another path.  For
those the stated tolerance is on single-Newton-step quantities right after initialisation: relative
1e-9 on the gradient norm, curvature and step lengths, direction error <= 1e-6 (maxDirectionError), and the solve must end
with status 0 and feasible outputs."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def libs():
    import __graft_entry__ as g
    return g.build()


def _load(so, B):
    from tenscalc_amd.runtime import HipSolver
    return HipSolver(so, B=B, device=0)


def _solve_both(spec, lib, data, args, B, tmp_path, tag):
    from tenscalc_amd.tapeio import save_tape
    from oracle.tc_oracle import Oracle
    tape = str(tmp_path / (tag + ".tape"))
    save_tape(spec.tape, tape)
    names = [n for n, _ in spec.parameters + spec.variables]
    stacked = {n: np.stack([np.asarray(d[n], dtype=float).reshape(-1, order="F") for d in data]) for n in names}
    obj = _load(lib, B)
    for n in names:
        obj.set(n, stacked[n])
    st, it, _ = obj.solve(**args)
    got = np.concatenate([a.reshape(B, -1) for a in obj.getOutputs()], axis=1)
    c = spec.classname
    ref, rst, rit = Oracle(tape).solve_batch({f"{c}_set_{n}": stacked[n] for n in names}, c + "_getOutputs", B,
                                             mu0=args["mu0"], maxIter=args["maxIter"])
    return (st, it, got), (rst, rit, ref), obj


def test_sls_interp_bit_exact(libs, sls_spec, tmp_path):
    from tenscalc_amd.problems.sls import SOLVE_ARGS, sls_data
    B = 200                                 # three full tiles + a ragged one
    data = []
    for k in range(B):
        A, b, x0 = sls_data(3000 + k)
        data.append(dict(A=A, b=b, x=x0))
    (st, it, got), (rst, rit, ref), _ = _solve_both(sls_spec, libs["sls_interp"], data, SOLVE_ARGS, B, tmp_path, "sls")
    assert np.array_equal(st, rst) and np.array_equal(it, rit) and (st == 0).all()
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("variant", ["sls_interp4", "sls_interp8", "sls_interp16"])
def test_sls_interp_cooperating_waves_bit_exact(libs, sls_spec, tmp_path, variant):
    """W wavefronts of a block share the 64 instances of a tile and split the records of every level
    (barrier records between levels): same bits as one wavefront per tile and as the oracle"""
    from tenscalc_amd.problems.sls import SOLVE_ARGS, sls_data
    B = 130
    data = []
    for k in range(B):
        A, b, x0 = sls_data(4000 + k)
        data.append(dict(A=A, b=b, x=x0))
    (st, it, got), (rst, rit, ref), _ = _solve_both(sls_spec, libs[variant], data, SOLVE_ARGS, B, tmp_path, variant)
    assert np.array_equal(st, rst) and np.array_equal(it, rit) and (st == 0).all()
    assert np.array_equal(got, ref)


def test_dcmotor_interp_bit_exact(libs, tmp_path):
    import __graft_entry__ as g
    from tenscalc_amd.problems.mpc_dcmotor import SOLVE_ARGS, dcmotor_data, mpc_dcmotor
    spec = mpc_dcmotor(folder=g.GENERATED)
    B = 130
    data = [dcmotor_data(4000 + k) for k in range(B)]
    (st, it, got), (rst, rit, ref), _ = _solve_both(spec, libs["dcmotor_interp"], data, SOLVE_ARGS, B, tmp_path, "dc")
    assert np.array_equal(st, rst) and np.array_equal(it, rit) and (st == 0).all()
    assert np.array_equal(got, ref)


def test_l1l2_interp_bit_exact(libs, tmp_path):
    """BASELINE configs[2]: 1792x1792 KKT, 54 k instructions, scaleCost"""
    import __graft_entry__ as g
    from tenscalc_amd.problems.l1l2 import SOLVE_ARGS, l1l2, l1l2_data
    spec = l1l2(folder=g.GENERATED)
    B = 6
    data = [l1l2_data(10 + k) for k in range(B)]
    (st, it, got), (rst, rit, ref), _ = _solve_both(spec, libs["l1l2_interp"], data, SOLVE_ARGS, B, tmp_path, "l1l2")
    assert np.array_equal(st, rst) and np.array_equal(it, rit) and (st == 0).all()
    assert np.array_equal(got, ref)


def test_small_quadcopter_interp(libs, tmp_path):
    """mpc_quadcopter with an 8-step horizon: sqrt-family opcodes, equality constraints, long dot
    products.  Tolerance-based (libm pow differs between device and host, see module docstring)."""
    import __graft_entry__ as g
    from tenscalc_amd.problems.mpc_quadcopter import SOLVE_ARGS, mpc_quadcopter, quadcopter_data
    from tenscalc_amd.tapeio import save_tape
    from oracle.tc_oracle import Oracle
    T8 = 8
    spec = mpc_quadcopter(T=T8, classname="tmp_quadcopter8", folder=g.GENERATED)
    B = 5
    data = [quadcopter_data(20 + k, T=T8) for k in range(B)]
    names = [n for n, _ in spec.parameters + spec.variables]
    obj = _load(libs["quad8_interp"], B)
    for n in names:
        obj.set(n, np.stack([np.asarray(d[n], dtype=float).reshape(-1, order="F") for d in data]))
    tape = str(tmp_path / "q8.tape")
    save_tape(spec.tape, tape)
    c = spec.classname
    mu0 = SOLVE_ARGS["mu0"]
    probes = ["getJ__", "getNorminf_Grad__", "getNorminf_G__", "getCurvature__", "getMaxAlphas_s__", "getDirectionError__"]
    ref = {p: [] for p in probes}
    for d in data:
        o = Oracle(tape)
        for n in names:
            o.set(f"{c}_set_{n}", d[n])
        o.set("setAddEye2HessianU__", [1e-9]); o.set("setAddEye2HessianEq__", [1e-9])
        o.copy("initPrimal__"); o.copy("scaleIneq__"); o.set("setMu__", [mu0]); o.copy("initDualIneq__"); o.copy("initDualEqX__")
        for p in probes:
            ref[p].append(o.get(p))
    for p in probes:
        got = obj.probe(p, mu0=mu0)
        r = np.stack(ref[p])
        if p == "getDirectionError__":
            assert (got <= 1e-6).all() and (r <= 1e-6).all()
        else:
            np.testing.assert_allclose(got, r, rtol=1e-9, atol=1e-12, err_msg=p)
    st, it, _ = obj.solve(**SOLVE_ARGS)
    assert (st == 0).all()
    out = obj.getOutputs(as_dict=True)
    m = out["m_thrust"].reshape(B, -1)
    assert (m >= 5 - 1e-6).all() and (m <= 20 + 1e-6).all()           # thrust limits (mpc_quadcopter.m:121-122)
