"""GPU parity tests proper: the HIP path, called through the C-ABI, against the CPU oracle on the
same seeded inputs, against the committed golden vectors, and -- at BASELINE's full batch --
through size-independent properties.

Parity classes (__graft_entry__.PARITY_CLASS).
`bit-exact` -- `strict` (one thread per instance) and `w8x` (8 lanes per instance, bench.py's default): the tape
is evaluated in the reference's operand order with IEEE fp64 +,-,*,/ (no FMA, ordered sums), so every output must be
BIT-IDENTICAL to the oracle, including status and iteration count, on every instance.
`solver-tolerance` -- `fast`, `w8`, `w8s`, ... (FMA contraction, reassociated sums, hoisted tape): not bit-identical; a
Newton line-search decision can flip on a 1e-17 residue (see tests/test_oracle_sls.py), so the stated tolerance is
the solver's own (desiredDualityGap=1e-5, gradTolerance=1e-4): same status on every instance, |iter difference| <= 2,
|J-J_ref| <= 5e-5, |x-x_ref| <= 5e-5, box-KKT residual <= 1e-4; the fraction of instances whose KKT residual is within
1e-10 of the oracle's (the north-star criterion) is computed over ALL instances, printed, and must stay above the
floor stated per variant -- these lowerings do NOT meet that criterion on every instance, which is why they are not
the benchmark default.
"""
import os

import numpy as np
import pytest

from tenscalc_amd.problems.sls import SOLVE_ARGS, sls_batch, sls_data

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sls_N100_n8.npz")


@pytest.fixture(scope="module")
def libs():
    import __graft_entry__ as g
    return g.build()


def _load(so, B):
    from tenscalc_amd.runtime import HipSolver
    return HipSolver(so, B=B, device=0)


def _set(obj, A, b, x0):
    obj.setP_A(np.ascontiguousarray(np.transpose(A, (0, 2, 1))))   # [B,n,N] C-order == MATLAB [N,n,B]
    obj.setP_b(b)
    obj.setV_x(x0)


def _oracle_batch(spec, tape_path, A, b, x0, **kw):
    from oracle.tc_oracle import Oracle
    o = Oracle(tape_path)
    c = spec.classname
    B = A.shape[0]
    args = dict(mu0=SOLVE_ARGS["mu0"], maxIter=SOLVE_ARGS["maxIter"])
    args.update(kw)
    return o.solve_batch({c + "_set_A": np.transpose(A, (0, 2, 1)).reshape(B, -1), c + "_set_b": b,
                          c + "_set_x": x0}, c + "_getOutputs", B, **args)


BIT_EXACT = ["sls_strict", "sls_w8x"]


@pytest.mark.parametrize("variant", BIT_EXACT)
def test_strict_bit_exact_vs_oracle(libs, sls_spec, sls_tape_path, variant):
    B = 1003                               # ragged: not a multiple of the 64-wide wavefront, nor of 8
    A, b, x0 = sls_batch(B, seed0=1000)
    obj = _load(libs[variant], B)
    _set(obj, A, b, x0)
    st, it, t = obj.solve(**SOLVE_ARGS)
    J, x = obj.getOutputs()
    ref, rst, rit = _oracle_batch(sls_spec, sls_tape_path, A, b, x0)
    assert np.array_equal(st, rst) and np.array_equal(it, rit)
    assert np.array_equal(J.reshape(B), ref[:, 0]) and np.array_equal(x.reshape(B, 8), ref[:, 1:])
    assert (st == 0).all()


@pytest.mark.parametrize("variant", BIT_EXACT)
def test_strict_golden_vectors(libs, variant):
    g = np.load(GOLD)
    B = len(g["seeds"])
    A, b, x0 = sls_batch(B, seed0=int(g["seeds"][0]))
    obj = _load(libs[variant], B)
    _set(obj, A, b, x0)
    st, it, _ = obj.solve(**SOLVE_ARGS)
    J, x = obj.getOutputs()
    assert np.array_equal(st, g["status"]) and np.array_equal(it, g["iter"])
    assert np.array_equal(J.reshape(B), g["J"]) and np.array_equal(x.reshape(B, 8), g["x"])


def test_strict_tape_probes_golden(libs):
    """tape evaluation alone (gradient, Hessian, KKT, LDL, solves) right after initialisation"""
    g = np.load(GOLD)
    B = len(g["seeds"])
    A, b, x0 = sls_batch(B, seed0=int(g["seeds"][0]))
    obj = _load(libs["sls_strict"], B)
    _set(obj, A, b, x0)
    for p in ["getJ__", "getNorminf_Grad__", "getGapMinFMinLambda__", "getCurvature__",
              "getDirectionError__", "getMaxAlphas_s__", "getGrad__"]:
        got = obj.probe(p, mu0=SOLVE_ARGS["mu0"])
        assert np.array_equal(got, g["probe_" + p].reshape(B, -1)), p


@pytest.mark.parametrize("variant", BIT_EXACT)
def test_strict_edge_cases_status_bits(libs, sls_spec, sls_tape_path, variant):
    """maxIter exhaustion (status 8|...), NaN parameters (status 4; for w8x: in one lane group of a wavefront, and
    a NaN that the reference's min/max chains SKIP because it is not their first operand), an infeasible start
    (status 1), zeros (the sign of a zero minimum is order dependent), negative curvature (second regularisation
    pass), B=1, other mu0/addEye."""
    B = 131
    A, b, x0 = sls_batch(B, seed0=5000)
    A[3] *= np.nan
    b[5] *= 1e6
    A[7] *= 0
    x0[9] = .05 - 1e-13
    x0[11] = .06                                  # outside the box
    x0[13, 3] = 0.0                               # on the boundary: min F == 0 exactly
    x0[15, 2] = np.nan                            # NaN in one component of the initial point only
    b[17, 40] = np.nan                            # NaN in one row of b only
    A[19] = 0; b[19] = 0                          # zero gradient: max_abs == 0
    for kw in [dict(mu0=1.0, maxIter=4), dict(mu0=1e-3, maxIter=30), dict(mu0=10.0, maxIter=60, addEye2Hessian=(1e-3, 1e-3)),
               dict(mu0=1.0, maxIter=30, addEye2Hessian=(-1.0, -1.0))]:
        obj = _load(libs[variant], B)
        _set(obj, A, b, x0)
        st, it, _ = obj.solve(**{**SOLVE_ARGS, **kw})
        J, x = obj.getOutputs()
        ref, rst, rit = _oracle_batch(sls_spec, sls_tape_path, A, b, x0, **kw)
        assert np.array_equal(st, rst) and np.array_equal(it, rit), kw
        assert np.array_equal(J.reshape(B), ref[:, 0], equal_nan=True)
        assert np.array_equal(x.reshape(B, 8), ref[:, 1:], equal_nan=True)
        assert st[3] == 4
    one = _load(libs[variant], 1)
    _set(one, A[:1], b[:1], x0[:1])
    st, it, _ = one.solve(**SOLVE_ARGS)
    ref, rst, rit = _oracle_batch(sls_spec, sls_tape_path, A[:1], b[:1], x0[:1])
    assert st[0] == rst[0] and it[0] == rit[0] and np.array_equal(one.getOutputs()[1].reshape(1, 8), ref[:, 1:])


@pytest.mark.parametrize("variant", ["sls_w8", "sls_w8s"])
def test_wave_cooperative_edge_cases(libs, sls_spec, sls_tape_path, variant):
    """the benchmarked lowerings on the inputs that exercise the status bits: NaN parameters (status 4 in that
    instance only -- its lane group must not disturb the 7 other instances of its wavefront), an infeasible start
    (status 1), maxIter exhaustion (status 8|...), an extreme scale, B not a multiple of 8 and B=1.  Status codes
    of these instances equal the oracle's; the ordinary ones agree within the stated tolerance."""
    B = 131
    A, b, x0 = sls_batch(B, seed0=5000)
    A[3] *= np.nan
    b[5] *= 1e6
    x0[9] = .06                                   # outside the box: min F <= 0 at the first test
    for kw in [dict(mu0=1.0, maxIter=30), dict(mu0=1.0, maxIter=4)]:
        obj = _load(libs[variant], B)
        _set(obj, A, b, x0)
        st, it, _ = obj.solve(**{**SOLVE_ARGS, **kw})
        J, x = obj.getOutputs()
        x = x.reshape(B, 8)
        ref, rst, rit = _oracle_batch(sls_spec, sls_tape_path, A, b, x0, **kw)
        assert st[3] == 4 == rst[3] and st[9] == rst[9] == 1
        assert np.array_equal(st, rst), kw
        ordinary = np.ones(B, dtype=bool)
        ordinary[[3, 5, 9]] = False
        assert np.abs(it - rit)[ordinary].max() <= 2
        assert np.abs(x - ref[:, 1:])[ordinary].max() <= 5e-5
    one = _load(libs[variant], 1)
    _set(one, A[:1], b[:1], x0[:1])
    st, it, _ = one.solve(**SOLVE_ARGS)
    ref, rst, rit = _oracle_batch(sls_spec, sls_tape_path, A[:1], b[:1], x0[:1])
    assert st[0] == rst[0] and abs(int(it[0]) - int(rit[0])) <= 2
    assert np.abs(one.getOutputs()[1].reshape(1, 8) - ref[:, 1:]).max() <= 5e-5


def test_reference_c_surface_single_instance(libs, sls_spec, sls_tape_path):
    """the reference's own C symbols (<cls>_set_*, ipmPD_CSsolver, <cls>_getOutputs) at B=1"""
    import ctypes
    L = ctypes.CDLL(libs["sls_strict"], mode=ctypes.RTLD_LOCAL)
    c = sls_spec.classname
    A, b, x0 = sls_data(11)
    dp = ctypes.POINTER(ctypes.c_double)
    Af = np.ascontiguousarray(A.reshape(-1, order="F"))
    getattr(L, c + "_set_A")(Af.ctypes.data_as(dp))
    getattr(L, c + "_set_b")(b.ctypes.data_as(dp))
    getattr(L, c + "_set_x")(x0.ctypes.data_as(dp))
    mu0, mi, si = ctypes.c_double(1.0), ctypes.c_int32(30), ctypes.c_int32(-1)
    ae = (ctypes.c_double * 2)(1e-9, 1e-9)
    st, it, tm = ctypes.c_int32(-1), ctypes.c_int32(-1), ctypes.c_double(0)
    L.ipmPD_CSsolver(ctypes.byref(mu0), ctypes.byref(mi), ctypes.byref(si), ae, ctypes.byref(st),
                     ctypes.byref(it), ctypes.byref(tm))
    J = ctypes.c_double(0)
    x = np.zeros(8)
    getattr(L, c + "_getOutputs")(ctypes.byref(J), x.ctypes.data_as(dp))
    ref, rst, rit = _oracle_batch(sls_spec, sls_tape_path, A[None], b[None], x0[None])
    assert (st.value, it.value) == (rst[0], rit[0])
    assert J.value == ref[0, 0] and np.array_equal(x, ref[0, 1:])


# measured on MI355X (round 2; printed by the test): fraction of instances with KKT residual within 1e-10 of the
# reference's -- floors a few points below the measurement
KKT_FLOOR = {"sls_fast": 0.78, "sls_w8": 0.75, "sls_w8h": 0.73, "sls_regs": 0.58, "sls_regsl": 0.58, "sls_w8s": 0.73}


@pytest.mark.parametrize("variant", ["sls_fast", "sls_w8", "sls_w8h", "sls_regs", "sls_regsl", "sls_w8s"])
def test_fast_build_within_stated_tolerance(libs, sls_spec, sls_tape_path, variant):
    """reassociating lowerings (fast, w8) and the hoisted tape (w8h, regs: tenscalc_amd/hoist.py) against
    the oracle running the REFERENCE tape"""
    B = 2048 + 5                            # ragged for the 8-instances-per-wave lowering too
    A, b, x0 = sls_batch(B, seed0=20000)
    obj = _load(libs[variant], B)
    _set(obj, A, b, x0)
    st, it, _ = obj.solve(**SOLVE_ARGS)
    J, x = obj.getOutputs()
    J, x = J.reshape(B), x.reshape(B, 8)
    ref, rst, rit = _oracle_batch(sls_spec, sls_tape_path, A, b, x0)
    assert np.array_equal(st, rst) and (st == 0).all()
    st_b, it_b, _ = obj.solve(**SOLVE_ARGS)                       # deterministic: no atomics, fixed reduction trees
    assert np.array_equal(it, it_b) and np.array_equal(obj.getOutputs()[1].reshape(B, 8), x)
    assert np.abs(it - rit).max() <= 2
    assert np.abs(J - ref[:, 0]).max() <= 5e-5
    assert np.abs(x - ref[:, 1:]).max() <= 5e-5
    assert (x >= 0).all() and (x <= .05).all()

    # KKT residual of a box-constrained problem, recomputed on the host in fp64:
    # r(x) = || x - clip(x - grad J(x), 0, .05) ||_inf
    def kkt(xv):
        gr = np.einsum("bij,bi->bj", A, np.einsum("bij,bj->bi", A, xv) - b) * (2 / 100)
        return np.abs(xv - np.clip(xv - gr, 0, .05)).max(axis=1)
    r_gpu, r_ref = kkt(x), kkt(ref[:, 1:])
    assert r_gpu.max() <= 1e-4 and r_ref.max() <= 1e-4          # both are solutions to solver tolerance
    # the north-star criterion, over ALL instances (no self-selected subset): fraction with KKT residual within
    # 1e-10 of the reference's.  These lowerings do not reach 1.0 (a flipped line-search branch moves the
    # returned point by ~1e-8..5e-6); the floor catches a regression, the printed number is the honest one
    within = np.abs(r_gpu - r_ref) <= 1e-10
    same_it = it == rit
    print("\n%s: |dKKT|<=1e-10 on %.4f of %d instances; same iteration count on %.4f; max|dKKT| %.3g, max|dx| %.3g"
          % (variant, within.mean(), B, same_it.mean(), np.abs(r_gpu - r_ref).max(), np.abs(x - ref[:, 1:]).max()))
    assert within.mean() >= KKT_FLOOR[variant], (variant, within.mean())
    assert np.abs(r_gpu - r_ref).max() <= 1e-5


def test_device_resident_inputs_both_layouts(libs, sls_spec, sls_tape_path):
    import torch
    B = 512
    A, b, x0 = sls_batch(B, seed0=777)
    ref, rst, rit = _oracle_batch(sls_spec, sls_tape_path, A, b, x0)
    from tenscalc_amd.runtime import LAYOUT_BATCH_MINOR, LAYOUT_INSTANCE_MAJOR
    Aim = torch.from_numpy(np.ascontiguousarray(np.transpose(A, (0, 2, 1)).reshape(B, 800))).cuda()
    bim = torch.from_numpy(b).cuda()
    xim = torch.from_numpy(x0).cuda()
    for layout in (LAYOUT_INSTANCE_MAJOR, LAYOUT_BATCH_MINOR):
        obj = _load(libs["sls_strict"], B)
        tr = (lambda t: t) if layout == LAYOUT_INSTANCE_MAJOR else (lambda t: t.t().contiguous())
        obj.set_device("A", tr(Aim), layout)
        obj.set_device("b", tr(bim), layout)
        obj.set_device("x", tr(xim), layout)
        obj.synchronize()
        st, it, _ = obj.solve(**SOLVE_ARGS)
        J, x = obj.getOutputs()
        assert np.array_equal(st, rst) and np.array_equal(it, rit)
        assert np.array_equal(x.reshape(B, 8), ref[:, 1:])


def test_caller_stream_is_ordered_with_the_handle_stream(libs, sls_spec, sls_tape_path):
    """include/tenscalc_hip.h "Devices and streams": inputs stored and a solve enqueued on a CALLER's stream,
    then getOutputs / set on the handle's own stream without any explicit synchronisation -- the results are those
    of the synchronous path (an event links the two streams; before, the transpose of getOutputs could read
    outputs the solve had not written yet)."""
    import torch
    from tenscalc_amd.runtime import LAYOUT_BATCH_MINOR
    B = 16384
    g_ = torch.Generator(device="cuda").manual_seed(7)
    A = torch.rand((800, B), generator=g_, device="cuda", dtype=torch.float64)
    b = torch.rand((100, B), generator=g_, device="cuda", dtype=torch.float64)
    x0 = .02 * torch.rand((8, B), generator=g_, device="cuda", dtype=torch.float64)
    ref = _load(libs["sls_w8x"], B)
    for nm, t in (("A", A), ("b", b), ("x", x0)):
        ref.set_device(nm, t, LAYOUT_BATCH_MINOR)
    st0, it0, _ = ref.solve(**SOLVE_ARGS)
    J0, x0_ = ref.getOutputs()
    side = torch.cuda.Stream()
    obj = _load(libs["sls_w8x"], B)
    for rep in range(3):
        with torch.cuda.stream(side):
            for nm, t in (("A", A), ("b", b), ("x", x0)):
                obj.set_device(nm, t, LAYOUT_BATCH_MINOR, stream=side.cuda_stream)
            obj.solve_async(SOLVE_ARGS["mu0"], SOLVE_ARGS["maxIter"], SOLVE_ARGS["addEye2Hessian"], stream=side.cuda_stream)
        J, x = obj.getOutputs()                                   # handle stream: must wait for the side stream
        assert np.array_equal(J, J0) and np.array_equal(x, x0_), rep
        obj.set("x", np.zeros((B, 8)))                            # handle stream again; the next round overwrites it
    obj.close(); ref.close()


def test_bench_default_bit_exact_on_bench_batch(libs, sls_spec, sls_tape_path):
    """bench.py's default variant on bench.py's own batch (B=65536, the generator of bench.make_batch): status,
    iteration count and outputs of 4096 + 1024 instances of that batch are bit-identical to the oracle, and the
    other bit-exact lowering (strict) agrees with it bit for bit on ALL 65536."""
    import torch
    import __graft_entry__ as g
    import bench
    from tenscalc_amd.runtime import LAYOUT_BATCH_MINOR
    assert g.PARITY_CLASS[g.BENCH_DEFAULT] == "bit-exact"
    B = 65536
    A, b, x0 = bench.make_batch(B, 0, torch.device("cuda", 0))
    res = {}
    for tag in ("sls_" + g.BENCH_DEFAULT, "sls_strict"):
        obj = _load(libs[tag], B)
        for nm, t in (("A", A), ("b", b), ("x", x0)):
            obj.set_device(nm, t, LAYOUT_BATCH_MINOR)
        obj.synchronize()
        st, it, _ = obj.solve(**SOLVE_ARGS)
        J, x = obj.getOutputs()
        res[tag] = (st, it, J.reshape(B), x.reshape(B, 8))
        obj.close()
    d, s_ = res["sls_" + g.BENCH_DEFAULT], res["sls_strict"]
    assert all(np.array_equal(a_, b_) for a_, b_ in zip(d, s_))
    assert (d[0] == 0).all()
    sel = np.concatenate([np.arange(4096), np.random.RandomState(3).choice(np.arange(4096, B), 1024, replace=False)])
    Ah = A[:, torch.from_numpy(sel).cuda()].t().cpu().numpy()           # [k][800]: column-major A of instance k
    bh = b[:, torch.from_numpy(sel).cuda()].t().cpu().numpy()
    xh = x0[:, torch.from_numpy(sel).cuda()].t().cpu().numpy()
    from oracle.tc_oracle import Oracle
    c = sls_spec.classname
    ref, rst, rit = Oracle(sls_tape_path).solve_batch({c + "_set_A": Ah, c + "_set_b": bh, c + "_set_x": xh},
                                                      c + "_getOutputs", len(sel), mu0=SOLVE_ARGS["mu0"],
                                                      maxIter=SOLVE_ARGS["maxIter"])
    assert np.array_equal(d[0][sel], rst) and np.array_equal(d[1][sel], rit)
    assert np.array_equal(d[2][sel], ref[:, 0]) and np.array_equal(d[3][sel], ref[:, 1:])


@pytest.mark.parametrize("variant", ["sls_strict", "sls_w8x", "sls_w8s"])
def test_full_batch_properties(libs, variant):
    """BASELINE config 2 size (B=65536): properties that need no CPU oracle -- every instance
    converges, respects the box, satisfies the KKT tolerances, is invariant to where it sits in the
    batch (permutation), and a second solve() reproduces the first bit-for-bit.  Run for the benchmark default
    (w8x), the other bit-exact lowering and the solver-tolerance variant bench.py offers as --variant w8s."""
    import torch
    B = 65536
    g = torch.Generator(device="cuda").manual_seed(1)
    A = torch.rand((800, B), generator=g, device="cuda", dtype=torch.float64)
    b = torch.rand((100, B), generator=g, device="cuda", dtype=torch.float64)
    x0 = .02 * torch.rand((8, B), generator=g, device="cuda", dtype=torch.float64)
    from tenscalc_amd.runtime import LAYOUT_BATCH_MINOR
    obj = _load(libs[variant], B)
    for nm, t in (("A", A), ("b", b), ("x", x0)):
        obj.set_device(nm, t, LAYOUT_BATCH_MINOR)
    obj.synchronize()
    st, it, t1 = obj.solve(**SOLVE_ARGS)
    J, x = obj.getOutputs()
    x = x.reshape(B, 8)
    assert (st == 0).all() and it.min() >= 8 and it.max() <= 20
    assert (x > 0).all() and (x < .05).all()
    Ah = A.cpu().numpy().reshape(8, 100, B)            # [n][N][B]
    r = np.einsum("jib,jb->ib", Ah, x.T) - b.cpu().numpy()
    assert np.allclose(J.reshape(B), (r ** 2).sum(0) / 100, rtol=1e-12)
    st2, it2, _ = obj.solve(**SOLVE_ARGS)
    J2, x2 = obj.getOutputs()
    assert np.array_equal(st, st2) and np.array_equal(it, it2) and np.array_equal(x2.reshape(B, 8), x)
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(2)).cuda()
    obj2 = _load(libs[variant], B)
    for nm, t in (("A", A), ("b", b), ("x", x0)):
        obj2.set_device(nm, t[:, perm].contiguous(), LAYOUT_BATCH_MINOR)
    obj2.synchronize()
    st3, it3, _ = obj2.solve(**SOLVE_ARGS)
    x3 = obj2.getOutputs()[1].reshape(B, 8)
    p = perm.cpu().numpy()
    assert np.array_equal(it3, it[p]) and np.array_equal(x3, x[p])


@pytest.mark.parametrize("tag", ["dcmotor_strict", "dcmotor_w32x"])
def test_dcmotor_strict_bit_exact_vs_oracle(libs, tmp_path, tag):
    """BASELINE configs[3] problem (equality + inequality constraints, 323x323 KKT, second LDL for
    initDualEqX__) through the generic one-thread-per-instance lowering and through the bit-exact wave-cooperative
    one at 32 lanes per instance (two instances per wavefront, state in LDS shared by life time: the faster of the
    two at one GPU's share of the sweep): bit-identical to the oracle."""
    import __graft_entry__ as g
    from tenscalc_amd.problems.mpc_dcmotor import SOLVE_ARGS as DC_ARGS, dcmotor_data, mpc_dcmotor
    from tenscalc_amd.tapeio import save_tape
    from oracle.tc_oracle import Oracle
    spec = mpc_dcmotor(folder=g.GENERATED)
    tape = str(tmp_path / "dc.tape")
    save_tape(spec.tape, tape)
    B = 200
    data = [dcmotor_data(1000 + k) for k in range(B)]
    obj = _load(libs[tag], B)
    names = [n for n, _ in spec.parameters + spec.variables]
    stacked = {n: np.stack([np.asarray(d[n], dtype=float).reshape(-1, order="F") for d in data]) for n in names}
    for n in names:
        obj.set(n, stacked[n])
    st, it, _ = obj.solve(**DC_ARGS)
    outs = obj.getOutputs()
    c = spec.classname
    o = Oracle(tape)
    ref, rst, rit = o.solve_batch({f"{c}_set_{n}": stacked[n] for n in names}, c + "_getOutputs", B,
                                  mu0=DC_ARGS["mu0"], maxIter=DC_ARGS["maxIter"])
    assert np.array_equal(st, rst) and np.array_equal(it, rit) and (st == 0).all()
    got = np.concatenate([a.reshape(B, -1) for a in outs], axis=1)
    assert np.array_equal(got, ref)


def test_odd_sized_sls_both_lowerings(libs, tmp_path):
    """sls with N=37 rows (not a multiple of the 8 lanes: zero padding + masked tail terms in the
    reductions) and n=5 unknowns: strict bit-exact, w8 within the stated tolerance; w8s = the hoisted tape with the
    inputs streamed (the last row block of the prologue sweep is ragged: 37 = 4*8+5)."""
    import __graft_entry__ as g
    from tenscalc_amd.problems.sls import sls
    from tenscalc_amd.tapeio import save_tape
    from oracle.tc_oracle import Oracle
    N, n, B = 37, 5, 777
    spec = sls(N=N, n=n, classname="tmp_sls_37_5", folder=g.GENERATED)
    tape = str(tmp_path / "s37.tape")
    save_tape(spec.tape, tape)
    rs = np.random.RandomState(5)
    A = rs.random_sample((B, N, n))
    b = rs.random_sample((B, N))
    x0 = .02 * rs.random_sample((B, n))
    At = np.ascontiguousarray(np.transpose(A, (0, 2, 1)))
    c = spec.classname
    ref, rst, rit = Oracle(tape).solve_batch({c + "_set_A": At.reshape(B, -1), c + "_set_b": b, c + "_set_x": x0},
                                             c + "_getOutputs", B, mu0=1.0, maxIter=30)
    for tag, exact in (("sls37_strict", True), ("sls37_w8x", True), ("sls37_w8", False), ("sls37_w8s", False)):
        obj = _load(libs[tag], B)
        obj.setP_A(At); obj.setP_b(b); obj.setV_x(x0)
        st, it, _ = obj.solve(**SOLVE_ARGS)
        J, x = obj.getOutputs()
        got = np.concatenate([J.reshape(B, 1), x.reshape(B, n)], axis=1)
        assert np.array_equal(st, rst)
        if exact:
            assert np.array_equal(it, rit) and np.array_equal(got, ref)
        else:
            assert np.abs(it - rit).max() <= 2 and np.abs(got - ref).max() <= 5e-5


def test_receding_horizon_warm_start_on_device(libs, tmp_path):
    """MPC loop of examples/mpcmhe/mpc_dcmotor.m:192-225 (setV_u(uWarm); setV_x(xWarm) from the previous
    outputs), three receding-horizon steps: the HIP path with the warm start fed back ON THE DEVICE
    (tc_set_from_output) is bit-identical, step by step, to the ORACLE running the same loop on the host (its
    getOutputs fed back through its sets), and to the HIP path with a host round trip."""
    import __graft_entry__ as g
    from tenscalc_amd.problems.mpc_dcmotor import SOLVE_ARGS as DC_ARGS, dcmotor_data, mpc_dcmotor
    from tenscalc_amd.tapeio import save_tape
    from oracle.tc_oracle import Oracle
    spec = mpc_dcmotor(folder=g.GENERATED)
    tape = str(tmp_path / "dc.tape")
    save_tape(spec.tape, tape)
    orc = Oracle(tape)
    c = spec.classname
    B = 64
    data = [dcmotor_data(2000 + k) for k in range(B)]
    names = [n for n, _ in spec.parameters + spec.variables]
    stacked = {n: np.stack([np.asarray(d[n], dtype=float).reshape(-1, order="F") for d in data]) for n in names}
    objs = [_load(libs["dcmotor_strict"], B) for _ in range(2)]
    for obj in objs:
        for n in names:
            obj.set(n, stacked[n])
    onames = [nm for nm, _ in spec.outputs]
    osize = [int(np.prod(sz)) if len(sz) else 1 for _, sz in spec.outputs]
    ooff = np.concatenate([[0], np.cumsum(osize)])
    cur = dict(stacked)                       # the oracle's inputs, step by step
    res = []
    for step in range(3):
        ref, rst, rit = orc.solve_batch({f"{c}_set_{n}": cur[n] for n in names}, c + "_getOutputs", B,
                                        mu0=DC_ARGS["mu0"], maxIter=DC_ARGS["maxIter"])
        rout = {nm: ref[:, ooff[k]:ooff[k + 1]] for k, nm in enumerate(onames)}
        st = []
        for which, obj in enumerate(objs):
            s_, it, _ = obj.solve(**DC_ARGS)
            out = obj.getOutputs(as_dict=True)
            st.append((s_, it, out["J"], out["x"]))
            assert np.array_equal(s_, rst) and np.array_equal(it, rit), (step, which)
            for nm in onames:
                assert np.array_equal(out[nm].reshape(B, -1), rout[nm]), (step, which, nm)
            if which == 0:      # device-side feedback
                obj.set_from_output("x", "xWarm")
                obj.set_from_output("u", "uWarm")
            else:               # host round trip
                obj.set("x", out["xWarm"])
                obj.set("u", out["uWarm"])
        assert all(np.array_equal(a, b) for a, b in zip(st[0], st[1])), step
        assert (st[0][0] == 0).all()
        res.append(st[0][2].copy())
        cur = dict(cur, x=rout["xWarm"].copy(), u=rout["uWarm"].copy())
    assert not np.array_equal(res[0], res[1])          # the warm start did change the problem data


# largest |dx| the solver-tolerance lowering w8 may show on the template-branch problems (measured round 2, printed by
# the test; "eq" runs with scaleCost, whose duality-gap test is ~40x looser)
W8_DX_TOL = {"eq": 1e-3, "eqonly": 1e-9, "free": 1e-9}


@pytest.mark.parametrize("kind", ["eq", "eqonly", "free"])
def test_template_branches_on_gpu(libs, tmp_path, kind):
    """device Newton loop under the other #if combinations (nG>0 + useInertia + uncoupled alphas +
    scaleCost; nF==0; unconstrained): strict bit-exact, w8 within tolerance."""
    import __graft_entry__ as g
    from tenscalc_amd.problems.small import small_ls
    from tenscalc_amd.tapeio import save_tape
    from oracle.tc_oracle import Oracle
    spec = small_ls(kind, folder=g.GENERATED)
    tape = str(tmp_path / "p.tape")
    save_tape(spec.tape, tape)
    B = 300
    rs = np.random.RandomState(11)
    A = rs.random_sample((B, 12, 4)); b = rs.random_sample((B, 12)); x0 = .02 * rs.random_sample((B, 4)) + .005
    At = np.ascontiguousarray(np.transpose(A, (0, 2, 1)))
    c = spec.classname
    ref, rst, rit = Oracle(tape).solve_batch({c + "_set_A": At.reshape(B, -1), c + "_set_b": b, c + "_set_x": x0},
                                             c + "_getOutputs", B, mu0=1e-2, maxIter=40)
    # strict_share: the bit-exact lowering with LDS/scratch slots shared by life time (soundness of the sharing)
    for tag, exact in (("small_%s_strict" % kind, True), ("small_%s_strict_share" % kind, True),
                       ("small_%s_w8x" % kind, True), ("small_%s_w8" % kind, False)):
        obj = _load(libs[tag], B)
        obj.setP_A(At); obj.setP_b(b); obj.setV_x(x0)
        st, it, _ = obj.solve(mu0=1e-2, maxIter=40)
        J, x = obj.getOutputs()
        got = np.concatenate([J.reshape(B, 1), x.reshape(B, 4)], axis=1)
        if exact:
            assert np.array_equal(st, rst) and np.array_equal(it, rit) and np.array_equal(got, ref, equal_nan=True)
        else:
            # solver-tolerance class: the same status on EVERY instance, the objective within the solver's
            # tolerance on every instance (scaleCost loosens the duality-gap test of kind "eq" by |10/J| ~ 40x, so a
            # flipped line-search branch can move weakly determined components of x further than J)
            d = np.abs(got - ref)
            same = np.array_equal(st, rst)
            print("\n%s: same status on %.4f of %d instances, max|dJ| %.3g, max|dx| %.3g, 80%% quantile %.3g"
                  % (tag, (st == rst).mean(), B, np.nanmax(d[:, 0]), np.nanmax(d[:, 1:]), np.nanquantile(d.max(axis=1), 0.8)))
            assert same
            assert np.nanmax(d[:, 0]) <= 1e-4 and np.nanquantile(d.max(axis=1), 0.8) <= 1e-6
            assert np.nanmax(d[:, 1:]) <= W8_DX_TOL[kind]


def test_slot_sharing_is_bit_exact(libs, sls_spec, sls_tape_path):
    """Lowering._assign_shared_slots (values of transient groups with disjoint life times share storage) under
    the bit-exact lowering of the REFERENCE tape: status, iteration counts and outputs identical to the
    oracle, which gives every value its own slot -- for converging, maxIter-limited and NaN instances."""
    B = 1500
    A, b, x0 = sls_batch(B, seed0=77000)
    A[3] *= np.nan
    x0[5] = .06                                   # infeasible start
    # addEye2Hessian < 0: negative curvature, the second pass of the regularisation loop runs (the only case in
    # which groups are re-evaluated inside one Newton iteration after a set)
    for maxIter, ae in ((30, (1e-9, 1e-9)), (4, (1e-9, 1e-9)), (30, (-10.0, -10.0)), (30, (-1.0, -1.0))):
        ref, rst, rit = _oracle_batch(sls_spec, sls_tape_path, A, b, x0, maxIter=maxIter, addEye2Hessian=ae)
        for tag in ("sls_strict_share", "sls_w8x"):        # w8x shares slots too (pushed groups keep their own)
            obj = _load(libs[tag], B)
            _set(obj, A, b, x0)
            args = dict(SOLVE_ARGS, maxIter=maxIter, addEye2Hessian=ae)
            st, it, _ = obj.solve(**args)
            J, x = obj.getOutputs()
            assert np.array_equal(st, rst) and np.array_equal(it, rit), tag
            assert np.array_equal(J.reshape(B), ref[:, 0], equal_nan=True), tag
            assert np.array_equal(x.reshape(B, 8), ref[:, 1:], equal_nan=True), tag
