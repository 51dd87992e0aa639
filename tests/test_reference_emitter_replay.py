"""This repo's tapes printed by the REFERENCE's own emitter and compared value by value with the oracle
(SURVEY.md 8c "hybrid oracle").

What runs: /root/reference/lib/csparse/instructionsTableUTHash.c is compiled where it lies (`-DIGNORE_MEX`) into a
scratch shared library; every instruction of a tape (types / parameters / operands as `Tape` holds them) is replayed
through the reference's `appendInstruction` (:406-444); the reference's `writeCinstructionsC` (:790-1609) prints the
instructions of a get's parent groups -- identity memory map, once with `minInstructions4Loop` = 50 (the reference's
default, parameters4all.m:438-447, so the rolled loops of :881-944 are exercised) and once with 10^9 (straight-line
code); the printed text is compiled with gcc -O1 (the reference's "fastest run-time" choice), the I_set slots (and the
values the printed groups read from other groups) are seeded from a finished oracle solve, and EVERY printed
instruction must equal the oracle's `tco_value` bit for bit.  The 17 static flop classes the reference returns
(:816-832) must equal `Tape.flops` summed over the same groups.

Standing of this test, stated plainly: the reference's file includes MATLAB's <mex.h>, which this image does not have;
the compile here is given an EMPTY file of that name and two typedefs on the command line (`mwSize`, `boolean_T`).  By
the task's rules a build that needs a stand-in header is not a "reference build" and pins nothing formally -- this
test therefore changes no parity label and nothing it builds is ever used by the oracle, the CPU baseline or the
product.  It is kept because it is the strongest evidence this container can produce that the per-opcode arithmetic
of oracle/tc_oracle.c (and of everything compared with it) is the arithmetic of the C the reference would print for
the same tape.  Skipped where /root/reference is absent (the GPU box)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

REF = "/root/reference/lib/csparse/instructionsTableUTHash.c"
pytestmark = pytest.mark.skipif(not os.path.isfile(REF), reason="/root/reference is not on this machine")

_i64p = ctypes.POINTER(ctypes.c_int64)
_dp = ctypes.POINTER(ctypes.c_double)


@pytest.fixture(scope="module")
def emitter(tmp_path_factory):
    d = tmp_path_factory.mktemp("ref_emitter")
    open(d / "mex.h", "w").close()                                  # see the module docstring: an empty file
    so = str(d / "libreftable.so")
    subprocess.run(["gcc", "-w", "-O1", "-shared", "-fPIC", "-DIGNORE_MEX", "-DmwSize=size_t", "-Dboolean_T=_Bool",
                    "-include", "stdbool.h", "-include", "stddef.h", "-I", str(d),
                    "-I", os.path.dirname(REF), REF, "-o", so, "-lm"], check=True)
    L = ctypes.CDLL(so)
    L.initInstructionsTable.restype = None
    L.appendInstruction.restype = ctypes.c_int64
    L.appendInstruction.argtypes = [ctypes.c_int, ctypes.c_int64, _dp, ctypes.c_int64, _i64p]
    L.writeCinstructionsC.restype = ctypes.c_int
    L.writeCinstructionsC.argtypes = [ctypes.c_char_p, _i64p, _i64p, _i64p, _i64p, ctypes.c_size_t, ctypes.c_size_t,
                                      ctypes.c_size_t]
    return L


def _replay(L, T):
    """the whole tape into the reference's table, ids preserved (appendInstruction returns 1-based indices)"""
    L.initInstructionsTable()
    pars = np.ascontiguousarray(T.pars, dtype=np.float64)
    ops = np.ascontiguousarray(T.ops, dtype=np.int64)
    pp = ctypes.cast(pars.ctypes.data, ctypes.c_void_p).value or 0
    op = ctypes.cast(ops.ctypes.data, ctypes.c_void_p).value or 0
    ps, os_ = T.parStart.tolist(), T.opStart.tolist()
    ty = T.types.tolist()
    for i in range(T.n):
        r = L.appendInstruction(ty[i], ps[i + 1] - ps[i], ctypes.cast(pp + 8 * ps[i], _dp),
                                os_[i + 1] - os_[i], ctypes.cast(op + 8 * os_[i], _i64p))
        assert r == i + 1, "the reference's table numbers the instructions differently"


_HARNESS = """
#include <math.h>
#include <float.h>
#include <stdint.h>
#include <string.h>
#define SCRATCHBOOK_TYPE double
static double m[%d];
%s
void run(const double *seed, double *out) {
  memcpy(m, seed, sizeof m);
%s
  memcpy(out, m, sizeof m);
}
"""


def _print_and_run(L, T, chunks, seed, workdir, tag, min4loop):
    """reference-printed C for `chunks` (lists of instruction ids in evaluation order; one printed file and one C function
    per chunk, as writeCfunctionpergroup.m:386-504 prints one function per dependency group) -> the scratchbook after
    running them on `seed`; the 17 flop counters the reference returned, summed; the printed text"""
    loc = np.arange(1, T.n + 1, dtype=np.int64)                  # identity map: instruction i lives in m[i-1]
    m4l = ctypes.c_int64(min4loop)
    total = np.zeros(17, dtype=np.int64)
    funs, calls, text = [], [], []
    for k, ids in enumerate(chunks):
        body = os.path.join(workdir, "%s_%d_%d.body.c" % (tag, min(min4loop, 99), k))
        idx = np.ascontiguousarray(ids, dtype=np.int64)
        cnt = np.zeros(17, dtype=np.int64)
        fn = body.encode()
        rc = L.writeCinstructionsC(fn, idx.ctypes.data_as(_i64p), loc.ctypes.data_as(_i64p), ctypes.byref(m4l),
                                   cnt.ctypes.data_as(_i64p), len(fn), idx.size, T.n)
        assert rc == 0
        total += cnt
        funs.append('static void f%d(void) {\n#include "%s"\n}' % (k, os.path.basename(body)))
        calls.append("  f%d();" % k)
        text.append(open(body).read())
    src = os.path.join(workdir, "%s_%d.c" % (tag, min(min4loop, 99)))
    with open(src, "w") as f:
        f.write(_HARNESS % (T.n, "\n".join(funs), "\n".join(calls)))
    so = src[:-2] + ".so"
    # -fno-builtin: at -O1 gcc fuses sin(x) and cos(x) of one operand into glibc's sincos(), whose cosine differs from
    # cos() in the last bit on 0.14 % of arguments (measured); the comparison is about the printed statements
    # the large tapes' straight-line text (one 20 000-statement function per factorisation) compiles at -O0: same statements
    opt = "-O0" if T.n > 20000 and min4loop > 50 else "-O1"
    subprocess.run(["gcc", "-w", opt, "-fno-builtin", "-ffp-contract=off", "-shared", "-fPIC", "-I", workdir, src, "-o", so,
                    "-lm"], check=True)
    R = ctypes.CDLL(so)
    out = np.empty(T.n)
    R.run(seed.ctypes.data_as(_dp), out.ctypes.data_as(_dp))
    return out, total, "".join(text)


def _check(L, o, T, gets, workdir, tag):
    """after a finished solve: evaluate every get on the oracle (no set in between, so every parent group ends up clean on
    the same inputs), print the union of their parent groups through the reference in the groups' order, compare"""
    from tenscalc_amd import instructions as I
    val = lambda i: o.L.tco_value(o.h, ctypes.c_int64(int(i)))      # noqa: E731
    groups = set()
    for name in gets:
        o.get(name)                                                   # parent groups evaluated, lazily, as in a solve
        groups.update(T.entry(name)[2]["parents"])
    groups = sorted(groups)                                           # group ids are a topological order (csparse.Tape)
    want = np.array([val(i) for i in range(1, T.n + 1)])
    loads = np.flatnonzero(T.types == I.I_load) + 1
    chunks = [loads] + [T.groups[g - 1] for g in groups]
    order = np.concatenate(chunks)
    seed = want.copy()
    seed[order - 1] = np.nan                                          # a value the printed code fails to write stays NaN
    compared, rolled = 0, 0
    for m4l in (50, 10 ** 9):
        got, cnt, text = _print_and_run(L, T, chunks, seed, workdir, tag, m4l)
        rolled += text.count("for(int64_t c=")
        sel = order - 1
        same = (got[sel].view(np.int64) == want[sel].view(np.int64)) | (np.isnan(got[sel]) & np.isnan(want[sel]))
        bad = order[~same]
        assert not bad.size, "%s: instruction(s) %s: reference-printed C %r, oracle %r" % (
            tag, bad[:5], got[bad[:5] - 1], want[bad[:5] - 1])
        assert np.array_equal(cnt, T.flops[groups].sum(axis=0)), (tag, cnt)
        compared += sel.size
    return compared, rolled


def _solve(o, spec, data, args):
    c = spec.classname
    for nm, _ in spec.parameters + spec.variables:
        o.set("%s_set_%s" % (c, nm), data[nm])
    return o.solve(args["mu0"], args["maxIter"], -1, args["addEye2Hessian"])


def _all_gets(T):
    return [g["name"] for g in T.gets]


def test_sls_tape_through_the_reference_emitter(emitter, native_built, sls_spec, tape_oracle, tmp_path):
    from tenscalc_amd.problems.sls import SOLVE_ARGS, sls_data
    T = sls_spec.tape
    _replay(emitter, T)
    o = tape_oracle(T)
    A, b, x0 = sls_data(3)
    st, it = _solve(o, sls_spec, dict(A=A, b=b, x=x0), SOLVE_ARGS)
    assert st == 0
    n, rolled = _check(emitter, o, T, _all_gets(T), str(tmp_path), "sls")
    assert n > 2 * 700 and rolled > 0            # the 100 x 8 and 8 x 100 families are printed as loops at the default threshold


def test_opcode_zoo_through_the_reference_emitter(emitter, native_built, tape_oracle, tmp_path):
    """every unary opcode of the ISA, I_componentwise included (printed from its parameter string, :1333-1345)"""
    from tenscalc_amd import instructions as I
    from tenscalc_amd.problems.small import opcode_zoo
    spec = opcode_zoo(folder=str(tmp_path))
    T = spec.tape
    _replay(emitter, T)
    o = tape_oracle(T)
    rs = np.random.RandomState(21)
    A = rs.random_sample((12, 4)); b = rs.random_sample(12); x0 = .02 * rs.random_sample(4) + .005
    st, it = _solve(o, spec, dict(A=A, b=b, x=x0), dict(mu0=1e-2, maxIter=40, addEye2Hessian=(1e-9, 1e-9)))
    assert st == 0
    n, _ = _check(emitter, o, T, _all_gets(T), str(tmp_path), "zoo")
    used = set(T.types[[i - 1 for g in T.groups for i in g]].tolist())
    assert n > 400 and len(used) >= 40 and I.I_componentwise in used


@pytest.mark.parametrize("config", ["dcmotor", "l1l2", "quadcopter"])
def test_config_tapes_through_the_reference_emitter(emitter, native_built, tape_oracle, tmp_path, config):
    if config == "dcmotor":
        from tenscalc_amd.problems.mpc_dcmotor import SOLVE_ARGS, dcmotor_data as data, mpc_dcmotor as make
    elif config == "l1l2":
        from tenscalc_amd.problems.l1l2 import SOLVE_ARGS, l1l2, l1l2_data as data
        make = l1l2
    else:
        from tenscalc_amd.problems.mpc_quadcopter import SOLVE_ARGS, mpc_quadcopter as make, quadcopter_data as data
    spec = make(folder=str(tmp_path))
    T = spec.tape
    _replay(emitter, T)
    o = tape_oracle(T)
    st, it = _solve(o, spec, data(2), SOLVE_ARGS)
    assert st == 0
    # the solver's own gets (exit tests, step lengths, direction error) and the user-facing outputs: between them every
    # group the Newton loop evaluates (KKT assembly, both factorisations, the solves) is a parent of one of them
    n, _ = _check(emitter, o, T, _all_gets(T), str(tmp_path), config)
    covered = set()
    for name in _all_gets(T):
        covered.update(T.entry(name)[2]["parents"])
    assert n >= 2 * sum(len(T.groups[g - 1]) for g in covered)
