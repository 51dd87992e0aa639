"""CPU: the record stream of the table-driven back-end (tenscalc_amd/interp_schedule.py), walked by a
Python model of csrc/tc_interp.hip.inc, reproduces the oracle's value of EVERY instruction bit for bit.

This pins the scheduler (levels, record packing, padding, row map, sign bits, long-instruction layout)
without a GPU; the device walk itself is checked against the oracle in tests/test_gpu_interp.py."""
import ctypes

import numpy as np
import pytest

from tenscalc_amd import instructions as I
from tenscalc_amd import interp_schedule as isch
from tenscalc_amd.lowering import Lowering

DOT_SKIP = {I.I_minus_dot: 0, I.I_minus_dot_div: 1, I.I_plus_minus_dot: 1, I.I_plus_minus_dot_div: 2}
f64 = np.float64


def _unary(op, x):
    with np.errstate(all="ignore"):
        return {
            I.I_inv: lambda: f64(1) / x, I.I_minus_inv_sqr: lambda: f64(-1) / (x * x),
            I.I_2_inv_cube: lambda: f64(2) / (x * x * x), I.I_abs: lambda: abs(x),
            I.I_sign: lambda: f64(1 if x >= 0 else -1), I.I_sqrt: lambda: np.sqrt(x),
            I.I_Dsqrt: lambda: f64(.5) * f64(float(x) ** -.5), I.I_DDsqrt: lambda: f64(-.25) * f64(float(x) ** -1.5),
            I.I_sqr: lambda: x * x, I.I_2times: lambda: f64(2) * x, I.I_2: lambda: f64(2),
            I.I_cube: lambda: x * x * x, I.I_3sqr: lambda: f64(3) * x * x, I.I_6times: lambda: f64(6) * x,
            I.I_relu: lambda: x if x > 0 else f64(0), I.I_heaviside: lambda: f64(1 if x > 0 else 0),
            isch.OP_COPY: lambda: x,
        }[op]()


def _eval(op, aux, v, sg):
    """one instruction, operands v (signs sg for I_sum), in the reference's evaluation order"""
    with np.errstate(all="ignore"):
        if op == I.I_sum:
            acc = -v[0] if sg[0] else v[0]
            for x, s in zip(v[1:], sg[1:]):
                acc = acc - x if s else acc + x
            return acc
        if op == I.I_sumprod:
            total = line = None
            nS = 0
            ns = len(v) // aux
            for s in range(ns):
                x = v[s * aux]
                for p in range(1, aux):
                    x = x * v[s * aux + p]
                line = x if line is None else line + x
                if s + 1 < ns:
                    nS += aux
                    if nS >= 100:
                        total = line if total is None else total + line
                        line, nS = None, 0
            return line if total is None else total + line
        if op in DOT_SKIP:
            sk = DOT_SKIP[op]
            acc = v[sk] * v[sk + 1]
            for k in range(sk + 2, len(v), 2):
                acc = acc + v[k] * v[k + 1]
            return {I.I_minus_dot: lambda: -acc, I.I_minus_dot_div: lambda: -acc / v[0],
                    I.I_plus_minus_dot: lambda: v[0] - acc, I.I_plus_minus_dot_div: lambda: (v[0] - acc) / v[1]}[op]()
        if op == I.I_div:
            return v[0] / v[1]
        if op in (I.I_plus_sqr, I.I_plus_abs):
            f = (lambda x: x * x) if op == I.I_plus_sqr else abs
            acc = f(v[0])
            for x in v[1:]:
                acc = acc + f(x)
            return acc
        if op in (I.I_min, I.I_max, I.I_min0, I.I_max0, I.I_max_abs):
            xs = [abs(x) for x in v] if op == I.I_max_abs else list(v)
            acc = f64(0) if op in (I.I_min0, I.I_max0) else xs.pop(0)
            for x in xs:
                if (acc > x) if op in (I.I_min, I.I_min0) else (acc < x):
                    acc = x
            return acc
        if op == I.I_clp:
            acc = f64(np.finfo(float).max)
            for k in range(0, len(v), 2):
                if v[k + 1] < 0:
                    t = -v[k] / v[k + 1]
                    if acc > t:
                        acc = t
            return acc
        return _unary(op, v[0])


def walk(words, ranges, m, base):
    """Python model of tc_interp(): m[base + row] is the state of one instance; ranges = one
    (start, end) per cooperating wave.  The waves run one at a time from barrier to barrier."""
    pcs = [s_ for s_, _ in ranges]
    while any(pc < e for pc, (_, e) in zip(pcs, ranges)):
        for k, (_, e) in enumerate(ranges):
            pcs[k] = _walk_to_barrier(words, pcs[k], e, m, base)
    assert all(pc == e for pc, (_, e) in zip(pcs, ranges))


def _walk_to_barrier(w, pc, end, m, base):
    while pc < end:
        op, K, n, aux, ln = w[pc] & 255, w[pc] >> 8, w[pc + 1], w[pc + 2], w[pc + 6]
        if K == 2:
            return pc + ln
        if K == isch.K_PAIR:          # 8 producer/consumer pairs: the producer's value goes straight into operand p
            opA, nA, auxA, p = w[pc + 3] & 255, (w[pc + 3] >> 8) & 255, w[pc + 3] >> 16, w[pc + 4]
            dest = w[pc + 8:pc + 16]
            ca = [w[pc + 16 + 8 * j:pc + 24 + 8 * j] for j in range(nA)]
            cb = [w[pc + 16 + 8 * (nA + j):pc + 24 + 8 * (nA + j)] for j in range(n)]
            va = [[m[base + (ca[j][k] >> 1)] for j in range(nA)] for k in range(8)]
            vb = [[m[base + (cb[j][k] >> 1)] if j != p else None for j in range(n)] for k in range(8)]
            for k in range(8):
                vb[k][p] = _eval(opA, auxA, va[k], [ca[j][k] & 1 for j in range(nA)])
            res = [_eval(op, aux, vb[k], [cb[j][k] & 1 for j in range(n)]) for k in range(8)]
            for k in range(8):
                m[base + dest[k]] = res[k]
        elif K == isch.K_LONG4:       # 4 long instructions of one shape, operands in chunks of 8 per instruction
            sk = DOT_SKIP.get(op, 0)
            nch = (n + 7) // 8
            res = []
            for q in range(4):
                a = [w[pc + 12 + 4 * h + q] for h in range(sk)]
                for c in range(nch):
                    a += w[pc + 24 + 32 * c + 8 * q:pc + 24 + 32 * c + 8 * (q + 1)]
                a = a[:sk + n]
                res.append(_eval(op, aux, [m[base + (x >> 1)] for x in a], [x & 1 for x in a]))
            for q in range(4):
                m[base + w[pc + 8 + q]] = res[q]
        elif K >= 8:
            dest = w[pc + 8:pc + 8 + K]
            cols = [w[pc + 8 + K + K * j:pc + 8 + K + K * (j + 1)] for j in range(n)]
            vals = [[m[base + (cols[j][k] >> 1)] for j in range(n)] for k in range(K)]      # all loads first
            res = [_eval(op, aux, vals[k], [cols[j][k] & 1 for j in range(n)]) for k in range(K)]
            for k in range(K):
                m[base + dest[k]] = res[k]
        elif K == 1:
            a = w[pc + 8:pc + 8 + n]
            m[base + w[pc + 3]] = _eval(op, aux, [m[base + (x >> 1)] for x in a], [x & 1 for x in a])
        else:
            sk = DOT_SKIP.get(op, 0)
            a = list(w[pc + 4:pc + 4 + sk]) + list(w[pc + 8:pc + 8 + n])
            m[base + w[pc + 3]] = _eval(op, aux, [m[base + (x >> 1)] for x in a], [x & 1 for x in a])
        pc += ln
    return pc


def _check(spec, data, tape_oracle, args, W=1):
    T = spec.tape
    c = spec.classname
    L = Lowering(T, None, roll=False, lanes=1, spill_all=True)
    cs = isch.class_stream(T, L, c, W)
    o = tape_oracle(T, c)
    for nm, _ in spec.parameters + spec.variables:
        o.set(f"{c}_set_{nm}", data[nm])
    D = spec.defines
    ae = args.get("addEye2Hessian", (1e-9, 1e-9))
    if D["setAddEye2Hessian"]:
        o.set("setAddEye2HessianU__", [ae[0]])
        o.set("setAddEye2HessianEq__", [ae[1]])
    o.copy("initPrimal__")
    if D["scaleCost"]:
        o.copy("scaleCost__")
    if D["nF"] > 0:
        if D["scaleInequalities"]:
            o.copy("scaleIneq__")
        o.set("setMu__", [args["mu0"]])
        o.copy("initDualIneq__")
    if D["nG"] > 0:
        o.copy("initDualEqX__")
    for e in T.gets:                       # evaluates every group a get depends on
        o.get(e["name"])
    val = lambda i: o.L.tco_value(o.h, ctypes.c_int64(int(i)))
    base = -cs["lo"]
    m = np.full(cs["hi"] - cs["lo"], np.nan)
    for i in range(1, T.n + 1):            # inputs, internal sets, constants
        if T.isset[i - 1] and (L.external[i] or L.pslot[i] >= 0):
            m[base + cs["row_of"](i)] = val(i)
    for k, v in enumerate(cs["consts"]):   # the kernel prologue: the tape's constants, then the rows of split reductions
        m[base + L.nPriv + k] = v
    assert all(cs["consts"][r - L.nPriv] == float(T.pars[T.parStart[i - 1]]) for i, r in cs["const_row"].items())
    needed = sorted({g for e in T.gets for g in e["parents"]})
    for g in needed:
        walk(cs["words"], cs["granges"][g - 1], m, base)
    bad = inside = 0
    for g in needed:
        for i in T.groups[g - 1]:
            a, b = m[base + cs["row_of"](i)], val(i)
            if a != a and b == b and L.single_local[i]:
                inside += 1                # evaluated inside its consumer's record, never stored
            elif not (a == b or (a != a and b != b)):
                bad += 1
    assert bad == 0, f"{bad} instruction values differ from the oracle"
    assert inside <= sum(cs["gstats"][g - 1][2] for g in needed)
    return cs


def test_stream_sls(sls_spec, tape_oracle):
    from tenscalc_amd.problems.sls import SOLVE_ARGS, sls_data
    A, b, x0 = sls_data(3)
    cs = _check(sls_spec, dict(A=A, b=b, x=x0), tape_oracle, SOLVE_ARGS)
    kinds = {cs["words"][pc] >> 8 for pc in _headers(cs)}
    assert {0, 1, 8, 16} <= kinds           # long, single, 8-wide and 16-wide records all occur


def _headers(cs):
    for rg in cs["granges"]:
        for s, e in rg:
            pc = s
            while pc < e:
                yield pc
                pc += cs["words"][pc + 6]


@pytest.mark.parametrize("W", [1, 4])
def test_stream_dcmotor(native_built, tape_oracle, tmp_path, W):
    """W=4: the records of every level spread over 4 cooperating waves, barrier records in between"""
    from tenscalc_amd.problems.mpc_dcmotor import SOLVE_ARGS, dcmotor_data, mpc_dcmotor
    spec = mpc_dcmotor(folder=str(tmp_path))
    cs = _check(spec, dcmotor_data(7), tape_oracle, SOLVE_ARGS, W)
    kinds = {cs["words"][pc] >> 8 for pc in _headers(cs)}
    assert {isch.K_PAIR, isch.K_LONG4} <= kinds      # producer/consumer pairs and 4-at-a-time long records occur
    assert sum(s_[2] for s_ in cs["gstats"]) > 1000  # ... pairs: more than a thousand producers never stored


def test_stream_small_quadcopter(native_built, tape_oracle, tmp_path):
    """sqrt / Dsqrt / DDsqrt / inv-family opcodes and long dot products (T=8 horizon)"""
    from tenscalc_amd.problems.mpc_quadcopter import SOLVE_ARGS, mpc_quadcopter, quadcopter_data
    spec = mpc_quadcopter(T=8, classname="tmp_quadcopter8", folder=str(tmp_path))
    _check(spec, quadcopter_data(1, T=8), tape_oracle, SOLVE_ARGS, W=8)


def test_validator_rejects_out_of_tile_rows(sls_spec):
    T = sls_spec.tape
    L = Lowering(T, None, roll=False, lanes=1, spill_all=True)
    cs = isch.class_stream(T, L, sls_spec.classname)
    w = list(cs["words"])
    pc = cs["granges"][0][0][0]
    w[pc + 8] = cs["hi"] * 2 + 100          # first operand/dest word of the first record
    with pytest.raises(AssertionError):
        isch.validate_stream(w, cs["granges"], cs["lo"], cs["hi"])


def test_validator_checks_pair_and_four_long_records(sls_spec):
    """every row of a pair record (both instructions' operands, except the producer's position, which must not name a row)
    and of a 4-long record (heads, every round's vectors) is checked on the host before the device walks the stream"""
    T = sls_spec.tape
    L = Lowering(T, None, roll=False, lanes=1, spill_all=True)
    cs = isch.class_stream(T, L, sls_spec.classname)
    first = {}
    for pc in _headers(cs):
        first.setdefault(cs["words"][pc] >> 8, pc)
    assert isch.K_PAIR in first and isch.K_LONG4 in first
    pc = first[isch.K_PAIR]
    nA, n, p = (cs["words"][pc + 3] >> 8) & 255, cs["words"][pc + 1], cs["words"][pc + 4]
    for off in (16 + 8 * nA - 1,                          # last operand word of the producers
                16 + 8 * (nA + p) + 3):                   # the producer's own position inside the consumer
        w = list(cs["words"])
        w[pc + off] = cs["hi"] * 2 + 64
        with pytest.raises(AssertionError):
            isch.validate_stream(w, cs["granges"], cs["lo"], cs["hi"])
    pc = first[isch.K_LONG4]
    for off in (9, 24 + 32 * ((cs["words"][pc + 1] + 7) // 8) - 1):    # a destination, the very last (padding) word
        w = list(cs["words"])
        w[pc + off] = cs["hi"] * 2 + 64
        with pytest.raises(AssertionError):
            isch.validate_stream(w, cs["granges"], cs["lo"], cs["hi"])
    w = list(cs["words"])
    w[pc + 6] += 8                                         # wrong length
    with pytest.raises(AssertionError):
        isch.validate_stream(w, cs["granges"], cs["lo"], cs["hi"])


def test_device_words_are_the_same_stream_in_byte_offsets(sls_spec):
    """what the device walks (interp_schedule.device_words): every dest / operand word of every record is its row's byte
    offset from the first row of the tile, sign bit kept; headers, the producer's position of a pair record and the output
    gather table are untouched -- decoded back it is the builder's stream word for word"""
    T = sls_spec.tape
    L = Lowering(T, None, roll=False, lanes=1, spill_all=True)
    for W in (1, 4):
        cs = isch.class_stream(T, L, sls_spec.classname, W)
        w, dv, lo = cs["words"], cs["dev_words"], cs["lo"]
        assert len(w) == len(dv)
        changed = [k for k in range(len(w)) if w[k] != dv[k]]
        assert changed
        rows_hi = (cs["hi"] - lo) * isch.ROW_BYTES
        nd = 0
        for k in changed:
            assert 0 <= dv[k] < rows_hi + 1 and (dv[k] & ~1) % isch.ROW_BYTES == 0
            as_dest, as_operand = dv[k] // isch.ROW_BYTES + lo, ((dv[k] >> 9) + lo) * 2 + (dv[k] & 1)
            assert w[k] in (as_dest, as_operand)
            nd += w[k] == as_dest
        assert nd > 0
        heads = set(_headers(cs))
        for pc in heads:                     # opcode / count / aux / length words never change
            for q in (0, 1, 2, 6, 7):
                assert dv[pc + q] == w[pc + q]
        s, n = cs["out_rng"]
        assert dv[s:s + n] == w[s:s + n]      # tc_gather reads rows


def test_shape_masks_cover_every_record_the_device_can_meet(native_built, tmp_path):
    """the per-class dispatcher (csrc/tc_interp.hip.inc under TC_OPS_*) instantiates only the shapes / opcodes of
    interp_schedule.shape_masks: every record of every range the device walks -- groups AND copy entry points -- must be
    in them, and the generated source must carry exactly those macros"""
    from tenscalc_amd import hip_emitter as he
    from tenscalc_amd.problems.mpc_dcmotor import mpc_dcmotor
    from tenscalc_amd import ldsm_schedule as lsch
    spec = mpc_dcmotor(folder=str(tmp_path))
    T = lsch.fused_tape(spec.tape, spec.defines)            # the groups the emitter walks (hip_emitter.emit_class)
    L = Lowering(T, None, roll=False, lanes=1, spill_all=True)
    cs = isch.class_stream(T, L, spec.classname, 4)
    m, w = cs["shapes"], cs["words"]
    seen = set()
    for rg in cs["granges"] + list(cs["copy_rng"].values()):
        for s, e in rg:
            pc = s
            while pc < e:
                op, K, n = w[pc] & 255, w[pc] >> 8, w[pc + 1]
                if K in (16, 8, 1):
                    assert (m["%d_%d" % (K, n)] >> op) & 1
                    seen.add("%d_%d" % (K, n))
                elif K == 0:
                    assert (m["long"] >> op) & 1
                elif K == isch.K_LONG4:
                    assert (m["long4"] >> op) & 1
                elif K == isch.K_PAIR:
                    assert (m["pairs"] >> ((((w[pc + 3] >> 8) & 255) - 1) * 3 + n - 1)) & 1
                pc += w[pc + 6]
    assert seen == {k for k in m if k[0].isdigit()}              # ... and no shape is listed that does not occur
    src = he.emit_class(spec, fp_contract="off", body="interp", waves=4)[0]
    for k, v in m.items():
        assert "#define TC_OPS_%s 0x%xULL\n" % (k, v) in src
    assert src.count("#define TC_OPS_") == len(m) + 1            # + TC_OPS_GIVEN


@pytest.mark.parametrize("op,aux", [(I.I_min, 0), (I.I_max, 0), (I.I_max_abs, 0), (I.I_clp, 0), (I.I_sumprod, 2), (I.I_sumprod, 3)])
def test_split_long_reductions_keep_the_reference_result(op, aux):
    """interp_schedule.split_long: a long reduction evaluated as parallel parts + one combining instruction returns the bits
    of the reference's chain -- also with NaNs in front / in the middle / at the head of a part, with +0 and -0 competing for
    a minimum, with infinities, with nothing to clip (DBL_MAX), and with the I_sumprod statement split every 100 factors"""
    rng = np.random.RandomState(5)

    def bits(x):
        return np.float64(x).view(np.int64)
    n = 1200 if op != I.I_sumprod else 996 * aux
    cases = []
    for trial in range(40):
        v = rng.standard_normal(n)
        if op == I.I_sumprod:
            v *= 10.0 ** rng.randint(-8, 8, n)           # sums whose order matters
        kind = trial % 8
        if kind == 1:
            v[0] = np.nan
        elif kind == 2:
            v[rng.randint(1, n, 5)] = np.nan
        elif kind == 3:                                   # NaN at the head of every later part, and the operands before it
            v[np.arange(32, n, 32)] = np.nan
            v[np.arange(31, n, 32)] = np.nan
        elif kind == 4:
            v[:] = np.where(rng.rand(n) < .5, 0.0, -0.0)
        elif kind == 5:
            v[rng.randint(0, n, 7)] = np.inf
            v[rng.randint(0, n, 7)] = -np.inf
        elif kind == 6 and op == I.I_clp:
            v[1::2] = np.abs(v[1::2])                     # no dx < 0: DBL_MAX
        elif kind == 7:
            v[:] = np.abs(v) + 1.0 if op in (I.I_min, I.I_clp) else -np.abs(v) - 1.0
        cases.append(v)
    for v in cases:
        rows = isch._ExtraRows(10 * n)
        a = [2 * k for k in range(n)]                     # operand k lives in row k
        sp = isch.split_long(op, aux, a, rows)
        assert sp is not None and len(sp[0]) >= 2
        m = {k: f64(x) for k, x in enumerate(v)}
        for k, c in enumerate(rows.vals):
            m[rows.base + k] = f64(c)
        whole = _eval(op, aux, [m[x >> 1] for x in a], [0] * n)
        for pop, paux, pa, pd in sp[0]:                   # the parts only read operands and constants: any order
            m[pd] = _eval(pop, paux, [m[x >> 1] for x in pa], [x & 1 for x in pa])
        fop, faux, fa = sp[1]
        got = _eval(fop, faux, [m[x >> 1] for x in fa], [x & 1 for x in fa])
        assert bits(got) == bits(whole) or (got != got and whole != whole), (I.NAMES[op], got, whole)
    assert isch.split_long(op, aux, [2 * k for k in range(64)], isch._ExtraRows(1000)) is None      # short ones are left alone


def test_stream_l1l2_splits_its_long_reductions(native_built, tape_oracle, tmp_path):
    """the l1l2 estimation problem at a horizon whose reductions are long enough to be split: parts + combining records in the
    stream, every instruction's value still the oracle's"""
    from tenscalc_amd.problems.l1l2 import SOLVE_ARGS, l1l2, l1l2_data
    spec = l1l2(N=140, classname="tmp_l1l2_140", folder=str(tmp_path))
    cs = _check(spec, l1l2_data(3, N=140), tape_oracle, SOLVE_ARGS, W=8)
    assert sum(s_[3] for s_ in cs["gstats"]) >= 4


def test_stream_long_reductions_with_nans(native_built, tape_oracle, tmp_path):
    """problems/small.py:long_reductions on the model of the walk (W = 8: every one of its five long reductions is split), operand
    lists with a NaN in front and at the head of every fetch chunk: each instruction's value is the oracle's (NaN where it is)"""
    from tenscalc_amd.problems.small import long_reductions
    spec = long_reductions(folder=str(tmp_path))
    rs = np.random.RandomState(2)
    v = rs.standard_normal(600); w = rs.standard_normal(600)
    v[0] = np.nan
    v[np.arange(32, 600, 32)] = np.nan
    v[5:9] = [0.0, -0.0, np.inf, -np.inf]
    data = dict(A=rs.random_sample((12, 4)), b=rs.random_sample(12), x=.02 * rs.random_sample(4) + .005, v=v, w=w)
    cs = _check(spec, data, tape_oracle, dict(mu0=1e-2, maxIter=40), W=8)
    assert sum(s_[3] for s_ in cs["gstats"]) == 5
