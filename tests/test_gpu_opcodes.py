"""GPU: the unary opcodes of the scalar ISA that the BASELINE problems do not reach (transcendental families,
rounding, derivative chains of sqrt / atan / srelu, I_componentwise) through every back-end, and a class built from
REFERENCE-dialect computation-graph files.

Tolerances, written here: opcodes made of IEEE + - * / sqrt fabs and comparisons are BIT-IDENTICAL to the oracle.
So are the opcodes that go through libm -- I_Dsqrt / I_DDsqrt (`.5*pow(x,-.5)`, `-.25*pow(x,-1.5)`) since round 4; exp, log,
the srelu family, sin, cos and atan since round 6: the device restates the host libm's routines operation by operation
(csrc/tc_glibcpow.hip.inc; tests/test_crmath.py compares them with the host's own on 10^7 arguments each).  No output of the
zoo has a tolerance any more.
The solver itself (status, iteration count, J, x) uses none of them and stays bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
LIBM = set()           # (rounds 1-5: the outputs built on exp / log / sin / cos / atan, compared within 4 ulp)


def _ulps(a, b):
    return np.abs(a - b) / np.spacing(np.maximum(np.abs(a), np.abs(b)))


@pytest.mark.parametrize("variant", ["strict", "w8x", "interp", "ldsm", "ldsc"])
def test_opcode_zoo(tmp_path, variant):
    import __graft_entry__ as g
    from tenscalc_amd.problems.small import opcode_zoo
    from tenscalc_amd.runtime import HipSolver
    from tenscalc_amd.tapeio import save_tape
    from oracle.tc_oracle import Oracle
    spec = opcode_zoo(folder=g.GENERATED)
    so = spec.build(**g.VARIANT_KW[variant])
    B = 200
    rs = np.random.RandomState(21)
    A = rs.random_sample((B, 12, 4)); b = rs.random_sample((B, 12)); x0 = .02 * rs.random_sample((B, 4)) + .005
    At = np.ascontiguousarray(np.transpose(A, (0, 2, 1)))
    tape = str(tmp_path / "zoo.tape")
    save_tape(spec.tape, tape)
    c = spec.classname
    ref, rst, rit = Oracle(tape).solve_batch({c + "_set_A": At.reshape(B, -1), c + "_set_b": b, c + "_set_x": x0},
                                             c + "_getOutputs", B, mu0=1e-2, maxIter=40)
    obj = HipSolver(so, B=B, device=0)
    obj.setP_A(At); obj.setP_b(b); obj.setV_x(x0)
    st, it, _ = obj.solve(mu0=1e-2, maxIter=40)
    out = obj.getOutputs(as_dict=True)
    assert np.array_equal(st, rst) and np.array_equal(it, rit) and (st == 0).all()
    off, worst = 0, {}
    for nm, sz in spec.outputs:
        k = int(np.prod(sz)) if len(sz) else 1
        got, want = out[nm].reshape(B, k), ref[:, off:off + k]
        off += k
        if nm in LIBM:
            u = _ulps(got, want).max()
            worst[nm] = u
            assert u <= 4, (nm, u)
        else:
            assert np.array_equal(got, want), nm
    print("\\n%s: worst ulp distance of the libm-based outputs:" % variant, {k: float(v) for k, v in worst.items()})
    assert len(worst) == len(LIBM)


def test_class_from_reference_dialect_cg_files(tmp_path, sls_spec):
    """tape -> files as the reference's saveScalarized writes them -> load_cg -> HIP class -> solve: the same bits as
    the oracle running the original tape (a MATLAB install hands any tape to the HIP back-end)"""
    from tenscalc_amd.cgformat import save_cg
    from tenscalc_amd.problems.sls import SOLVE_ARGS, sls_batch
    from tenscalc_amd.runtime import HipSolver
    from tenscalc_amd.tapeio import save_tape
    from tenscalc_amd.tools.build_class import spec_from_cg
    from oracle.tc_oracle import Oracle
    base = str(tmp_path / "slsref")
    save_cg(sls_spec.tape, base, word=32, extensions="opcodes")
    spec2 = spec_from_cg(base, defines=sls_spec.defines, folder=str(tmp_path))
    assert spec2.classname == sls_spec.classname and [nm for nm, _ in spec2.parameters] == ["A", "b", "x"]
    so = spec2.build(fp_contract="off", tag="fromcg", static_clean=True)
    B = 300
    A, b, x0 = sls_batch(B, seed0=8100)
    obj = HipSolver(so, B=B, device=0)
    obj.set("A", np.ascontiguousarray(np.transpose(A, (0, 2, 1))))
    obj.set("b", b)
    obj.set("x", x0)
    st, it, _ = obj.solve(**SOLVE_ARGS)
    got = np.concatenate([a.reshape(B, -1) for a in obj.getOutputs()], axis=1)
    tape = str(tmp_path / "orig.tape")
    save_tape(sls_spec.tape, tape)
    c = sls_spec.classname
    ref, rst, rit = Oracle(tape).solve_batch({c + "_set_A": np.transpose(A, (0, 2, 1)).reshape(B, -1), c + "_set_b": b,
                                              c + "_set_x": x0}, c + "_getOutputs", B, mu0=SOLVE_ARGS["mu0"],
                                             maxIter=SOLVE_ARGS["maxIter"])
    assert np.array_equal(st, rst) and np.array_equal(it, rit) and np.array_equal(got, ref)


def _adversarial(B, M, rs):
    """operand lists for long reductions: ordinary values, NaN in front, NaNs inside, a NaN at the head of every 32-chunk (where a
    part of a split reduction may start) and behind it, +0 / -0 only, infinities, nothing to clip, all of one sign"""
    v = rs.standard_normal((B, M)) * 10.0 ** rs.randint(-6, 6, (B, M))
    w = rs.standard_normal((B, M))
    for k in range(B):
        kind = k % 8
        if kind == 1:
            v[k, 0] = np.nan
        elif kind == 2:
            v[k, rs.randint(1, M, 5)] = np.nan
        elif kind == 3:
            v[k, np.arange(32, M, 32)] = np.nan
            v[k, np.arange(31, M, 32)] = np.nan
        elif kind == 4:
            v[k] = np.where(rs.rand(M) < .5, 0.0, -0.0)
        elif kind == 5:
            v[k, rs.randint(0, M, 7)] = np.inf
            v[k, rs.randint(0, M, 7)] = -np.inf
        elif kind == 6:
            w[k] = np.abs(w[k])                      # no dx < 0: clp returns DBL_MAX
        elif kind == 7:
            v[k] = np.abs(v[k]) + 1.0
    return v, w


@pytest.mark.parametrize("variant", ["interp", "interp8", "ldsc4"])
def test_long_reductions_with_nans_zeros_infinities(tmp_path, variant):
    """the long minimum / maximum / infinity norm / clp / inner product of 600 operands -- on `interp8` evaluated as parallel
    parts + a combining record (interp_schedule.split_long), on the LDS walk as two-level folds -- return the bits of the
    reference's chain for operand lists with NaNs (in front, inside, at the head of every chunk), +0 / -0, infinities"""
    import __graft_entry__ as g
    from tenscalc_amd.problems.small import long_reductions
    from tenscalc_amd.runtime import HipSolver
    from tenscalc_amd.tapeio import save_tape
    from oracle.tc_oracle import Oracle
    spec = long_reductions(folder=g.GENERATED)
    so = spec.build(**g.VARIANT_KW[variant])
    if variant == "interp8":
        src = open(so.replace("lib" + spec.classname, spec.classname).replace(".so", ".hip")).read()
        assert "__builtin_inf()" in src          # the constants the parts of a split minimum / maximum lead with
    B, M = 128, 600
    rs = np.random.RandomState(33)
    A = rs.random_sample((B, 12, 4)); b = rs.random_sample((B, 12)); x0 = .02 * rs.random_sample((B, 4)) + .005
    At = np.ascontiguousarray(np.transpose(A, (0, 2, 1)))
    v, w = _adversarial(B, M, rs)
    tape = str(tmp_path / "lr.tape")
    save_tape(spec.tape, tape)
    c = spec.classname
    ref, rst, rit = Oracle(tape).solve_batch({c + "_set_A": At.reshape(B, -1), c + "_set_b": b, c + "_set_x": x0,
                                              c + "_set_v": v, c + "_set_w": w}, c + "_getOutputs", B, mu0=1e-2, maxIter=40)
    obj = HipSolver(so, B=B, device=0)
    obj.setP_A(At); obj.setP_b(b); obj.setV_x(x0); obj.setP_v(v); obj.setP_w(w)
    st, it, _ = obj.solve(mu0=1e-2, maxIter=40)
    got = np.concatenate([a.reshape(B, -1) for a in obj.getOutputs()], axis=1)
    obj.close()
    assert np.array_equal(st, rst) and np.array_equal(it, rit) and (st == 0).all()
    assert got.shape == ref.shape
    nan = np.isnan(ref)
    assert np.array_equal(np.isnan(got), nan)
    assert np.array_equal(got[~nan].view(np.int64), ref[~nan].view(np.int64))        # signs of zeros included
    assert nan[:, -5:].any() and (ref[:, -5:][~nan[:, -5:]] == 0).any()               # ... and the cases were there
