"""The one table of numbers the reference itself holds for this path: /root/reference/examples/flops.m:29-42 -- operations
per solve (read from the generated library's own profile, flops.m:93-95, @csparse/writeCprofiling.c:16-22) and
`code.statistics.sizeScratchbook` (flops.m:121) of `J=x*A*x-b*x` through cmex2optimizeCS for N = 30 .. 300.

One integer per N is the product of the front-end (tprod / gradient / hessian, flattened sums), every scalarizer on the
path (tprod, plus, ldl, mldivide, norminf), the CSE of the symmetric Hessian, the dependency groups with their lazy flags,
the Newton loop's call sequence and the per-opcode accounting of writeCinstructionsC.  The table is reproduced EXACTLY
with `adjustAddEye2Hessian=False` -- driven by the reference's own ipmPD_CSsolver.c compiled in place, by the oracle's
restated loop, and (-m gpu) by the device loop through tc_profiling_view."""
import os

import numpy as np
import pytest

from tenscalc_amd import tcalculus as tc
from tenscalc_amd.problems.flops import FLOPS_PER_SOLVE, SCRATCHBOOK, SOLVE_ARGS, flops_batch, flops_data, flops_problem
from tenscalc_amd.tapeio import save_tape

SIZES = sorted(FLOPS_PER_SOLVE)


def closed_form(N):
    """(2N^3 + 69N^2 + 55N - 24) / 6: the cubic through the five exactly known entries of the table (it also gives
    10 037 746 at N=300, which prints as the table's 1.003775e+07).  Term by term, for one solve = two passes of the loop
    head, one Newton step: LDL' of the dense N x N matrix N^3/3 + N^2 - 4N/3 (sparsity_ldl.m:161-222: N(N-1)/2 products
    l*d, per column a k-term I_plus_minus_dot and N-col k-term I_plus_minus_dot_div); Hessian A+A'+eps*I N(N-1)/2 + 2N
    (symmetric entries share one I_sum, csparse.m:1168-1190); two gradient evaluations 2*(2*(2N^2-N) + 2N + (3N-2));
    right-hand side 2N; forward + diagonal + backward solves 2N^2 - N; primal update 2N."""
    v = 2 * N ** 3 + 69 * N ** 2 + 55 * N - 24
    assert v % 6 == 0
    return v // 6


def test_table_is_the_closed_form():
    for N in SIZES:
        assert float("%.6e" % closed_form(N)) == FLOPS_PER_SOLVE[N]
        assert SCRATCHBOOK[N] == 2 * N * N + 11 * N + 8


def _solve_counting(o, cls, A, b, x0):
    o.reset_counters()
    o.set(cls + "_set_A", A.reshape(-1, order="F"))
    o.set(cls + "_set_b", b)
    o.set(cls + "_set_x", x0)
    st, it = o.solve(SOLVE_ARGS["mu0"], SOLVE_ARGS["maxIter"], -1, SOLVE_ARGS["addEye2Hessian"])      # flops.m:84-88: no getOutputs
    return st, it, o.counters()["flops"]


@pytest.mark.parametrize("N", SIZES)
def test_flops_per_solve_equal_the_reference_table(N, tmp_path):
    from oracle.tc_oracle import Oracle, ReferenceLoopOracle
    tc.clear()
    spec = flops_problem(N, adjustAddEye2Hessian=False)
    tape = str(tmp_path / "flops.tape")
    save_tape(spec.tape, tape)
    A, b, x0 = flops_data(0, N)
    oracles = [Oracle(tape)]
    if os.path.isdir("/root/reference"):          # the reference's own Newton loop, compiled where it lies (oracle/build.py)
        oracles.append(ReferenceLoopOracle(tape, "flops%d_noadjust" % N, spec.defines))
    for o in oracles:
        for rep in range(2):        # flops.m:80-88 repeats the sequence: the second pass must count what the first did
            st, it, fl = _solve_counting(o, spec.classname, A, b, x0)
            assert (st, it) == (0, 2), "one Newton step, clean exit at the second pass of the loop head"
            assert fl == closed_form(N)
            assert float("%.6e" % fl) == FLOPS_PER_SOLVE[N]
        x = o.get(spec.classname + "_getOutputs")[1:]
        assert np.abs(np.linalg.solve(A + A.T, b) - x).max() <= 1e-7       # flops.m:92: x = inv(A+A')*b


@pytest.mark.parametrize("N", [30, 60, 100])
def test_todays_default_executes_one_more_hessian_vector_product(N, tmp_path):
    """with adjustAddEye2Hessian=true (parameters4optimize.m:116-117, today's default) the loop also calls
    getCurvature__ and getDirectionError__ (ipmPD_CSsolver.c:407,414): WW*dx (2N^2-N), dU'(WW dU) (2N-1), WW*dx-b (3N:
    the nested sum is flattened to four terms, Tcalculus.m:2283-2296) and its norminf (3N-2)"""
    from oracle.tc_oracle import Oracle
    tc.clear()
    spec = flops_problem(N)
    assert spec.defines["adjustAddEye2Hessian"] == 1
    tape = str(tmp_path / "flops.tape")
    save_tape(spec.tape, tape)
    st, it, fl = _solve_counting(Oracle(tape), spec.classname, *flops_data(1, N))
    assert (st, it) == (0, 2)
    assert fl - closed_form(N) == 2 * N * N + 7 * N - 3


def _without_gets(names):
    from tenscalc_amd import csparse
    orig = csparse.CSparse.declareGet

    def declare(self, sources, name, output_names=None):
        if name not in names:
            return orig(self, sources, name, output_names)
    return csparse.CSparse, "declareGet", declare


@pytest.mark.parametrize("N", SIZES)
def test_scratchbook_size_against_the_reference_table(N, monkeypatch):
    """dependencyGroups.m:275-422 restated (Tape.reference_memory_map) on this repo's tape.  The quadratic term of the
    table -- N^2 for A, N(N+1)/2 distinct Hessian entries, N(N-1)/2 for L, and the N(N-1)/2 products l*d of the
    factorisation squeezed into N-1 locations by the within-group reuse -- is reproduced; the linear term is not:
    today's assembly declares getHessInertia__, getDirectionError__ and getCurvature__ (ipmPD_CStimesLambda.m:218-219,
    230,241), which the flop column proves the table's solver never called.  With them 5N+4 locations more than the
    table, without them N+1."""
    tc.clear()
    T = flops_problem(N).tape
    loc, size = T.reference_memory_map()
    assert size == loc.max() + 1 and (loc[1:] >= 0).all()
    assert size - SCRATCHBOOK[N] == 5 * N + 4
    assert T.n - size == (N * (N - 1) // 2 - (N - 1)) + (N - 1) + (2 * N - 1)       # reuses: l*d, solves, inertia
    tc.clear()
    monkeypatch.setattr(*_without_gets({"getHessInertia__", "getDirectionError__", "getCurvature__"}))
    _, size = flops_problem(N).tape.reference_memory_map()
    assert size - SCRATCHBOOK[N] == N + 1


@pytest.mark.parametrize("N", [5, 12, 30])
def test_scratchbook_size_term_by_term(N, monkeypatch):
    """Where every location of 2N^2 + 12N + 9 (= the table's 2N^2 + 11N + 8, plus N + 1) comes from, dependency group by
    dependency group, under the option set that reproduces the flop column (the three gets the table's solver never
    called left out).  sizeScratchbook = instructions - reuses (dependencyGroups.m:277: one location per instruction,
    sets and constants included; :357-375 within-group reuse; :405-422 compression), so it depends on the ORDER in which
    the generator of the day created its instructions as well as on their number:

      N^2 + 3N + 5   sets: A (N^2), b, the user's x and the packed variable x_ (cmex2optimizeCS.m:209-218: packVariables
                     + the copy initPrimal__), and five scalars -- addEye2HessianU/Eq (ipmPD_CStimesLambda.m:107-110),
                     alphaPrimal / alphaDualEq / alphaDualIneq (:226-231, declared whatever nG and nF are)
      1              the constant 1 of Teye(size(Lf_uu)) (:235), which no instruction reads once eps*1 is folded
      2N             A*x and A'*x (the two tprods of the gradient), shared by getJ__ / getGrad__
      N(N-1)/2       off-diagonal Hessian entries A(i,j)+A(j,i), one instruction per symmetric pair (csparse.m:1168-1190)
      2              x*(A*x) and b*x;  1: their difference J;  N: the gradient;  1: its norminf
      N(N-1)/2 + 2N  the factorisation group: L (N(N-1)/2, read by the solves: never reused), D (N-1; D(1) is the
                     diagonal entry itself), the N diagonal sums H(i,i)+eps, and ONE location for all N(N-1)/2 products
                     l*d (sparsity_ldl.m:161-222: column c's products die with the column, and take the locations of
                     diagonal sums that are dead by then)
      N              the right-hand side -Lf_u;  2N - 1: forward, diagonal and backward solves (dx is read outside the group,
                     y and z reuse);  2N: alphaPrimal*dU and u + alphaPrimal*dU

    The table has N + 1 fewer.  Its generator is not in /root/reference (the flop column proves it predates three of
    today's entry points; the tree has no history), and the count depends on the creation order of instructions in the
    MATLAB front-end of that day, so the residual cannot be derived from the sources that exist: candidates of exactly
    that size in TODAY's assembly are one of the N-vectors above (the packed copy x_ of a single optimisation variable;
    the separate -Lf_u) together with one scalar set or the unread constant.  The test pins what CAN be pinned: every term
    of today's count."""
    tc.clear()
    monkeypatch.setattr(*_without_gets({"getHessInertia__", "getDirectionError__", "getCurvature__"}))
    T = flops_problem(N, adjustAddEye2Hessian=False).tape
    loc, size = T.reference_memory_map()
    per = {}
    for g in range(T.nGroups + 1):
        ids = np.flatnonzero(T.group[1:] == g) + 1 if g else np.flatnonzero(T.group[1:] == 0) + 1
        per[g] = (len(ids), len(set(loc[ids].tolist())))
    h = N * (N - 1) // 2
    want = [(N * N + 3 * N + 5 + 1,) * 2, (N, N), (N, N), (h, h), (1, 1), (1, 1), (2 * h + 2 * N - 1, h + 2 * N),
            (1, 1), (N, N), (1, 1), (N, N), (3 * N - 2, 2 * N - 1), (2 * N, 2 * N)]
    assert [per[g] for g in range(T.nGroups + 1)] == want
    assert sum(w[1] for w in want) == size == 2 * N * N + 12 * N + 9 == SCRATCHBOOK.get(N, 2 * N * N + 11 * N + 8) + N + 1


def test_reference_memory_map_is_a_valid_allocation():
    """no two values that are live at the same time share a location: replaying the tape in instruction order, every
    operand still holds the value its producer wrote"""
    tc.clear()
    T = flops_problem(12).tape
    loc, size = T.reference_memory_map()
    holder = -np.ones(size, dtype=np.int64)
    for i in range(1, T.n + 1):
        for j in T.ops[T.opStart[i - 1]:T.opStart[i]]:
            assert holder[loc[j]] == j, "instruction %d reads %d after its location was reused" % (i, j)
        holder[loc[i]] = i


@pytest.mark.gpu
@pytest.mark.parametrize("N", [30, 60])
def test_device_profile_total_equals_the_reference_table(N, tmp_path):
    """the same integer from the DEVICE: a stats=True class (the reference's profiling=true), one batched solve, and the
    token after the first "Total" of tc_profiling_view's file -- read exactly as flops.m:93-95 reads the reference's"""
    import __graft_entry__ as g
    from tenscalc_amd.runtime import HipSolver
    tc.clear()
    spec = flops_problem(N, adjustAddEye2Hessian=False, folder=g.GENERATED)
    B = 192
    A, b, x0 = flops_batch(B, seed0=7, N=N)
    obj = HipSolver(spec.build(**dict(g.VARIANT_KW["strict"], stats=True, tag="strict_stats")), B=B)
    obj.set("A", np.ascontiguousarray(np.transpose(A, (0, 2, 1))))
    obj.set("b", b)
    obj.set("x", x0)
    st, it, _ = obj.solve(**SOLVE_ARGS)
    assert (st == 0).all() and (it == 2).all()
    path = str(tmp_path / "flops.profile")
    obj.profilingView(path)
    prof = open(path).read().split()
    total = float(prof[prof.index("Total") + 1])
    assert total == closed_form(N) and float("%.6e" % total) == FLOPS_PER_SOLVE[N]
    assert obj.static_count("reference_scratchbook") - SCRATCHBOOK[N] == 5 * N + 4
    J, x = obj.getOutputs()
    for k in (0, B - 1):
        assert np.abs(np.linalg.solve(A[k] + A[k].T, b[k]) - x[k]).max() <= 1e-7
    obj.close()
