"""CPU: the record stream of the LDS-resident tape machine (tenscalc_amd/ldsm_schedule.py), walked by a Python
model of csrc/tc_ldsm.hip.inc along the entry-point sequence of the Newton loop, reproduces the oracle's value
of every `get` bit for bit.

This pins the scheduler without a GPU: the storage map (LDS / HBM split, blocks sharing slots by life time, the
group-local scratch region), the levels, the record packing, the split of long instructions into parallel terms
+ an ordered fold, sign bits, heads of the dot opcodes.  Storage is SHARED by life time, so the model has to
call the entry points in the order the solver does (csrc/ipm_newton.hip.inc); the device walk itself is checked
against the oracle in tests/test_gpu_interp.py."""
import os

import numpy as np
import pytest

from tenscalc_amd import instructions as I
from tenscalc_amd import ldsm_schedule as lsch
from test_interp_stream import _eval

f64 = np.float64
FOLD_TERM = {lsch.F_SUM: I.I_sum}


def _fold(op, aux, xs, h0, h1):
    """K_FOLD record: xs = the staged terms, left to right"""
    with np.errstate(all="ignore"):
        if op in lsch._DOT_SKIP:
            acc = xs[0]
            for x in xs[1:]:
                acc = acc + x
            return {I.I_minus_dot: lambda: -acc, I.I_minus_dot_div: lambda: -acc / h0,
                    I.I_plus_minus_dot: lambda: h0 - acc, I.I_plus_minus_dot_div: lambda: (h0 - acc) / h1}[op]()
        if op == I.I_sumprod:
            total = line = None
            nS = 0
            for s, x in enumerate(xs):
                line = x if line is None else line + x
                if s + 1 < len(xs):
                    nS += aux
                    if nS >= 100:
                        total = line if total is None else total + line
                        line, nS = None, 0
            return line if total is None else total + line
        if op == I.I_clp:
            acc = f64(np.finfo(float).max)
            for x in xs:
                if acc > x:
                    acc = x
            return acc
        if op in (lsch.F_MININF, lsch.F_MAXINF):       # one chunk of a long minimum / maximum: from +-inf
            acc = f64(np.inf if op == lsch.F_MININF else -np.inf)
            for x in xs:
                if (acc > x) if op == lsch.F_MININF else (acc < x):
                    acc = x
            return acc
        if op == lsch.F_SUM:
            acc = xs[0]
            for x in xs[1:]:
                acc = acc + x
            return acc
        return _eval(op, 0, xs, [0] * len(xs))        # I_min, I_min0, I_max, I_max0


def _term(op, aux, v, sg):
    if op == lsch.OP_PROD3:
        with np.errstate(all="ignore"):
            return v[0] * (v[1] * v[2])
    if op == lsch.OP_PRODDN:
        with np.errstate(all="ignore"):
            return (v[0] / v[2]) * ((v[1] / v[2]) * v[2])
    if op == lsch.OP_CLPTERM:
        with np.errstate(all="ignore"):
            return -v[0] / v[1] if v[1] < 0 else f64(np.finfo(float).max)
    return _eval(op, aux, v, sg)


class Machine:
    """one instance: m[address] (LDS and HBM part in one array), lazy group flags, the record walk"""

    def __init__(self, T, cs):
        self.T, self.cs = T, cs
        self.m = np.full(cs["nLds"] + cs["nGlob"], np.nan)
        for a, v in cs["consts"]:
            self.m[a] = v
        self.m[cs["dummy"]] = 0.0
        self.clean = set()
        self.hd = np.asarray(cs["headers"], dtype=np.int64).reshape(-1, 8)
        self.words = cs["words"]
        self.W = cs["W"]

    def run(self, rid):
        tb = self.cs["table"][rid * 2 * self.W:(rid + 1) * 2 * self.W]
        pcs = [tb[2 * k] for k in range(self.W)]
        ends = [tb[2 * k + 1] for k in range(self.W)]
        while any(p < e for p, e in zip(pcs, ends)):       # the waves run one at a time from barrier to barrier
            for k in range(self.W):
                pcs[k] = self._to_barrier(pcs[k], ends[k])

    def _to_barrier(self, h, end):
        m, w = self.m, self.words
        while h < end:
            w0, n, aux, cnt, off, flags, cn, _ = (int(x) for x in self.hd[h])
            kind, op = w0 >> 8, w0 & 255
            h += 1
            if kind == lsch.K_BARRIER:
                return h
            body = lsch.record_body(w, self.hd[h - 1])
            rows = lambda j: [int(x) for x in body[j]]                # noqa: E731
            dst = rows(0)
            res = []
            if kind == lsch.K_SHORT:
                cols = [rows(1 + j) for j in range(n)]
                for q in range(cnt):                                   # all operand reads of the record first
                    ws = [cols[j][q] for j in range(n)]
                    res.append(_term(op, aux, [m[x >> 1] for x in ws], [x & 1 for x in ws]))
            elif kind == lsch.K_FOLD:
                ln, h0, h1, base = rows(1), rows(2), rows(3), rows(4)
                llast = bool(aux & lsch.LATE_LAST) and op in lsch._DOT_SKIP    # late terms behind the chain, not in the row
                late = [[] for _ in range(cnt)]
                if (flags >> lsch.F_NLATE_SHIFT) & 3:            # late terms: computed by the folding lane itself
                    nl = rows(5)
                    for q in range(cnt):
                        for t in range(nl[q]):                  # a * (b * c); c = the slot holding 1.0 for a plain product
                            pos, wa, wb, wc = (rows(6 + 4 * t + j)[q] for j in range(4))
                            with np.errstate(all="ignore"):
                                xa, xb, xc = m[wa >> 1], m[wb >> 1], m[wc >> 1]
                                if op in lsch._DOT_SKIP and (aux >> (lsch.DN_SHIFT + t)) & 1:      # OP_PRODDN form
                                    val = (xa / xc) * ((xb / xc) * xc)
                                else:
                                    val = xa * (xb * xc)
                                if llast:
                                    late[q].append(val)
                                else:
                                    m[base[q] + pos] = val
                for q in range(cnt):
                    # a padded row (ordered sums) is added up to the next multiple of 8: the pads must hold -0.0
                    nrow = -(-ln[q] // 8) * 8 if flags & lsch.F_PADDED else ln[q]
                    res.append(_fold(op, (aux & ~lsch.LATE_LAST if llast else aux) & ~(3 << lsch.DN_SHIFT) if op in lsch._DOT_SKIP else aux,
                                     [m[base[q] + k] for k in range(nrow)] + late[q], m[h0[q] >> 1], m[h1[q] >> 1]))
            else:
                ln = rows(1)
                for q in range(cnt):
                    ws = [rows(2 + j)[q] for j in range(2)][:lsch._DOT_SKIP.get(op, 0)] + \
                         [rows(4 + k)[q] for k in range(ln[q])]
                    res.append(_eval(op, aux, [m[x >> 1] for x in ws], [x & 1 for x in ws]))
            for q in range(cnt):
                m[dst[q]] = res[q]
            if flags & lsch.F_BARRIER_AFTER:                         # the step ends behind this record
                return h
        return h

    # entry points (writeCfunctionpergroup.m:513-852)
    def need(self, parents):
        for g in parents:
            if g not in self.clean:
                self.clean.add(g)
                self.run(self.cs["rid"][g])

    def set(self, name, values):
        _, _, e = self.T.entry(name)
        for i, v in zip(e["ins"], np.asarray(values, dtype=float).reshape(-1, order="F")):
            if self.cs["addr"][i] >= 0:
                self.m[self.cs["addr"][i]] = v
        self.clean -= set(e["children"])

    def get(self, name):
        _, _, e = self.T.entry(name)
        self.need(e["parents"])
        out = []
        for o in e["outputs"]:
            for i in o["ins"]:
                out.append(float(self.T.pars[self.T.parStart[i - 1]]) if self.T.isload[i - 1] else self.m[self.cs["addr"][i]])
        return np.array(out)

    def copy(self, name):
        _, _, e = self.T.entry(name)
        self.need(e["parents"])
        self.run(self.cs["rid"][name])
        self.clean -= set(e["children"])


def _check(spec, data, tape_oracle, args, W=1, lds_slots=20000, stream_kw=None, fuse=False):
    T, c, D = spec.tape, spec.classname, spec.defines
    o = tape_oracle(T, c)                    # the oracle always walks the reference's partition into groups
    if fuse in ("loop", "loop+split"):       # ... the machine the coarser one of hip_emitter.emit_class_ldsm(fuse_chain=True):
        T = lsch.fused_tape(T, D, split_div=fuse == "loop+split")      # factor + solves, then the groups one solve cannot tell
        assert T.nGroups < spec.tape.nGroups      # apart (+split: the long (h0 - sum) / D(c) as numerator + division, TC_LDSM_SPLIT_DIV=1)
    elif fuse:                               # (only the first step)
        fs = lsch.fusion_sets(T, min_gain=32 if fuse is True else int(fuse))
        assert fs, "nothing to fuse on this tape"
        T = T.merged(fs)
    cs = lsch.class_stream(T, c, D, W=W, lds_slots=lds_slots, **(stream_kw or {}))
    mach = Machine(T, cs)

    def both(kind, name, *a):
        if kind == "get":
            got, ref = mach.get(name), o.get(name)
            assert got.shape == ref.shape and all(x == y or (x != x and y != y) for x, y in zip(got, ref)), \
                f"{name}: {np.sum(got != ref)} of {ref.size} values differ from the oracle"
            return ref
        getattr(mach, kind)(name, *a)
        getattr(o, kind)(name, *a)
    for nm, _ in spec.parameters + spec.variables:
        both("set", f"{c}_set_{nm}", data[nm])
    ae = args.get("addEye2Hessian", (1e-9, 1e-9))
    if D["setAddEye2Hessian"]:
        both("set", "setAddEye2HessianU__", [ae[0]])
        both("set", "setAddEye2HessianEq__", [ae[1]])
    both("copy", "initPrimal__")
    if D["scaleCost"]:
        both("copy", "scaleCost__")
        both("get", "getScale4Cost__")
    mu = args["mu0"]
    if D["nF"] > 0:
        if D["scaleInequalities"]:
            both("copy", "scaleIneq__")
        both("set", "setMu__", [mu])
        both("copy", "initDualIneq__")
    if D["nG"] > 0:
        both("copy", "initDualEqX__")
    for it in range(2):                      # two passes of the loop body of csrc/ipm_newton.hip.inc
        both("get", "getNorminf_Grad__")
        if D["nG"] > 0:
            both("get", "getNorminf_G__")
        if D["nF"] > 0:
            both("get", "getGapMinFMinLambda__")
        both("get", "getCurvature__")
        both("get", "getDirectionError__")
        if D["nF"] > 0:
            al = both("get", "getMaxAlphas_s__")
            a0 = .99 * min(1.0, float(np.min(al)))
            for trial in (a0, a0 * .99):
                both("set", "setAlphaPrimal__", [trial])
                both("get", "getMinF_s__")
            if D["nG"] > 0:
                both("set", "setAlphaDualEq__", [a0])
            both("set", "setAlphaDualIneq__", [a0])
        else:
            both("set", "setAlphaPrimal__", [1.0])
            if D["nG"] > 0:
                both("set", "setAlphaDualEq__", [1.0])
        both("copy", "updatePrimalDual__")
        if D["nF"] > 0:
            mu *= .2
            both("set", "setMu__", [mu])
    both("get", f"{c}_getOutputs")
    return cs


def test_ldsm_stream_sls(sls_spec, tape_oracle):
    from tenscalc_amd.problems.sls import SOLVE_ARGS, sls_data
    A, b, x0 = sls_data(3)
    cs = _check(sls_spec, dict(A=A, b=b, x=x0), tape_oracle, SOLVE_ARGS)
    kinds = set((np.asarray(cs["headers"]).reshape(-1, 8)[:, 0] >> 8).tolist())
    assert {lsch.K_SHORT, lsch.K_FOLD} <= kinds            # the 300-operand sums are staged and folded


@pytest.mark.parametrize("W", [1, 4, 8])
def test_ldsm_stream_dcmotor(native_built, tape_oracle, tmp_path, W):
    """nG > 0 (second LDL for initDualEqX__), W = 4 / 8: records of every step spread over the waves, the barrier
    riding on each wave's last record of the step"""
    from tenscalc_amd.problems.mpc_dcmotor import SOLVE_ARGS, dcmotor_data, mpc_dcmotor
    spec = mpc_dcmotor(folder=str(tmp_path))
    _check(spec, dcmotor_data(7), tape_oracle, SOLVE_ARGS, W)


@pytest.mark.parametrize("W", [1, 4])
def test_ldsm_stream_dcmotor_fused_factor_and_solves(native_built, tape_oracle, tmp_path, W):
    """the KKT factorisation and the triangular solves walked as ONE level schedule (csparse.Tape.merged +
    ldsm_schedule.fusion_sets): coarser dependency groups, fewer barrier-separated steps, the same values"""
    from tenscalc_amd.problems.mpc_dcmotor import SOLVE_ARGS, dcmotor_data, mpc_dcmotor
    spec = mpc_dcmotor(folder=str(tmp_path))
    T = spec.tape
    fs = lsch.fusion_sets(T)
    T2 = T.merged(fs)
    assert T2.nGroups == T.nGroups - sum(len(s_) - 1 for s_ in fs)
    assert sum(lsch.stream_levels(T2)[1]) <= sum(lsch.stream_levels(T)[1]) - 32 * len(fs)
    with pytest.raises(ValueError, match="cycle"):
        a, b_ = fs[0][0], fs[0][-1]
        if len(fs[0]) > 2:
            T.merged([[a, b_]])
        else:
            raise ValueError("cycle: (no group lies between these two)")
    _check(spec, dcmotor_data(7), tape_oracle, SOLVE_ARGS, W, fuse=True)


def test_ldsm_stream_small_quadcopter_fused(native_built, tape_oracle, tmp_path):
    from tenscalc_amd.problems.mpc_quadcopter import SOLVE_ARGS, mpc_quadcopter, quadcopter_data
    spec = mpc_quadcopter(T=8, classname="tmp_quadcopter8", folder=str(tmp_path))
    _check(spec, quadcopter_data(1, T=8), tape_oracle, SOLVE_ARGS, W=4, fuse=4)      # short horizon: small gains count


def test_ldsm_stream_dcmotor_small_lds(native_built, tape_oracle, tmp_path):
    """a tight LDS budget forces blocks into the per-instance HBM array: cold addresses, global flags"""
    from tenscalc_amd.problems.mpc_dcmotor import SOLVE_ARGS, dcmotor_data, mpc_dcmotor
    spec = mpc_dcmotor(folder=str(tmp_path))
    cs = _check(spec, dcmotor_data(8), tape_oracle, SOLVE_ARGS, W=2, lds_slots=4600)
    assert cs["nCold"] > 0 and cs["nLds"] <= 4600


def test_ldsm_stream_small_quadcopter(native_built, tape_oracle, tmp_path):
    """sqrt / Dsqrt / DDsqrt / inv-family opcodes and long dot products (T=8 horizon)"""
    from tenscalc_amd.problems.mpc_quadcopter import SOLVE_ARGS, mpc_quadcopter, quadcopter_data
    spec = mpc_quadcopter(T=8, classname="tmp_quadcopter8", folder=str(tmp_path))
    _check(spec, quadcopter_data(1, T=8), tape_oracle, SOLVE_ARGS, W=4)


def test_ldsm_validator_rejects_bad_addresses(sls_spec):
    cs = lsch.class_stream(sls_spec.tape, sls_spec.classname, sls_spec.defines)
    w = list(cs["words"])
    hd = np.asarray(cs["headers"]).reshape(-1, 8)
    w[int(hd[0, 4])] = cs["nLds"] + cs["nGlob"] + 5        # destination of the first record outside the state
    with pytest.raises(AssertionError):
        lsch.validate(w, cs["headers"], cs["table"], cs["W"], cs["nLds"], cs["nLds"] + cs["nGlob"])


def test_ldsm_validator_rejects_a_cross_wave_hazard_inside_a_step():
    """the waves of a step run side by side on the device: a record of one wave must not write what a record of
    another wave reads or writes before the next barrier.  The schedules built for W=4 pass; redirecting one
    destination onto an operand of another wave's record of the same step is caught (a sequential model that runs
    the waves one after the other would compute SOME value and say nothing)."""
    from tenscalc_amd.problems.mpc_dcmotor import mpc_dcmotor
    spec = mpc_dcmotor(folder="/tmp/ldsm_hazard_gen")
    cs = lsch.class_stream(spec.tape, spec.classname, spec.defines, W=4)
    W, hdrs, table = cs["W"], np.asarray(cs["headers"]).reshape(-1, 8), cs["table"]
    nA = cs["nLds"] + cs["nGlob"]
    dead = {cs["rid"][g_] for g_ in cs["lowering"].dead_groups if g_ in cs["rid"]}
    lsch.validate(cs["words"], cs["headers"], table, W, cs["nLds"], nA, dummy=cs["dummy"], unwalked=dead)   # as built: clean
    # find a range in which wave 0 and wave 1 both have a SHORT record in their first step
    for r0 in range(0, len(table), 2 * W):
        if r0 // (2 * W) in dead:
            continue
        h0, h1 = table[r0], table[r0 + 2]
        if table[r0 + 1] > h0 and table[r0 + 3] > h1 and hdrs[h0, 0] >> 8 == lsch.K_SHORT and hdrs[h1, 0] >> 8 == lsch.K_SHORT:
            rd1, _ = lsch.record_accesses(cs["words"], hdrs[h1])
            victim = next(a for a in sorted(rd1) if a != cs["dummy"] and a < cs["nLds"])
            w = list(cs["words"])
            w[int(hdrs[h0, 4])] = victim                   # wave 0's first destination := an operand of wave 1
            with pytest.raises(AssertionError, match="writes what wave"):
                lsch.validate(w, cs["headers"], table, W, cs["nLds"], nA, dummy=cs["dummy"], unwalked=dead)
            return
    pytest.skip("no range with two parallel short records")


@pytest.mark.parametrize("op", [I.I_min, I.I_min0, I.I_max, I.I_max0, I.I_clp])
def test_two_level_fold_equals_the_reference_chain(op):
    """ldsm_schedule._SPLIT: a long minimum / maximum folded chunk by chunk (chunks from +-inf, the reference's leading
    operand in front of the chunk results where its chain starts from it) gives the value of the reference's single
    chain bit for bit -- NaNs (sticky in first position, skipped elsewhere), signed zeros (the first of equal zeros
    stays), infinities and DBL_MAX included"""
    rng = np.random.RandomState(7)
    fop1, lead = lsch._SPLIT[op]
    special = [np.nan, 0.0, -0.0, np.inf, -np.inf, np.finfo(float).max, -np.finfo(float).max, 1.0, -1.0]
    for trial in range(400):
        n = int(rng.randint(97, 260))
        x = rng.randn(n)
        for _ in range(int(rng.randint(0, 12))):
            x[rng.randint(n)] = special[rng.randint(len(special))]
        if trial % 5 == 0:
            x[0] = special[trial // 5 % len(special)]
        if trial % 7 == 0:
            x[:] = np.where(rng.rand(n) < .5, 0.0, -0.0)
        xs = [f64(v) for v in x]
        ref = _fold(op, 0, xs, f64(0), f64(0))
        parts = [_fold(fop1, 0, xs[k:k + lsch.SPLIT_CHUNK], f64(0), f64(0)) for k in range(0, n, lsch.SPLIT_CHUNK)]
        got = _fold(op, 0, ([xs[0]] if lead else []) + parts, f64(0), f64(0))
        assert (got != got and ref != ref) or (got == ref and np.signbit(got) == np.signbit(ref)), (trial, got, ref)


@pytest.mark.parametrize("kw", [dict(hoist=False), dict(nested=False), dict(split=False), dict()])
def test_ldsm_stream_schedule_options_agree(native_built, tape_oracle, tmp_path, kw):
    """every scheduling step of DESIGN.md 3.9 -- terms staged one step ahead + late terms in the folding lane (hoist),
    nested products (nested), two-level folds of the long reductions (split) -- can be switched off, and every
    combination walks to the oracle's values: the steps change WHEN an operation runs, never what it computes"""
    from tenscalc_amd.problems.mpc_dcmotor import SOLVE_ARGS, dcmotor_data, mpc_dcmotor
    spec = mpc_dcmotor(folder=str(tmp_path))
    cs = _check(spec, dcmotor_data(9), tape_oracle, SOLVE_ARGS, W=4, stream_kw=kw)
    hd = np.asarray(cs["headers"]).reshape(-1, 8)
    steps = int(((hd[:, 0] >> 8) == lsch.K_BARRIER).sum() + (((hd[:, 0] >> 8) != lsch.K_BARRIER) & ((hd[:, 5] & lsch.F_BARRIER_AFTER) > 0)).sum())
    print(kw, "records", len(hd), "barriers (all waves)", steps)


def test_late_terms_behind_the_chain_are_used_and_can_be_switched_off(native_built, tape_oracle, tmp_path, monkeypatch):
    """ldsm_schedule.LATE_LAST: rows whose late terms are their last terms keep -0.0 in the late slots and add the late
    products behind the chain -- the model above executes exactly that against the oracle; TC_LDSM_LATE_LAST=0 gives
    the round-3 stream (late terms written into the row), same values"""
    from tenscalc_amd.problems.mpc_dcmotor import SOLVE_ARGS, dcmotor_data, mpc_dcmotor
    spec = mpc_dcmotor(folder=str(tmp_path))
    cs = _check(spec, dcmotor_data(9), tape_oracle, SOLVE_ARGS, W=4, fuse=True)
    hd = np.asarray(cs["headers"], dtype=np.int64).reshape(-1, 8)
    f = hd[(hd[:, 0] >> 8) == lsch.K_FOLD]
    late = ((f[:, 5] >> lsch.F_NLATE_SHIFT) & 3) > 0
    assert (late & ((f[:, 2] & lsch.LATE_LAST) != 0)).sum() > 50
    for h in f[late & ((f[:, 2] & lsch.LATE_LAST) != 0)]:       # such a record never writes into its rows
        rd, wr = lsch.record_accesses(cs["words"], h)
        body = lsch.record_body(cs["words"], h)
        assert wr == set(int(x) for x in body[0][:int(h[3])])
    monkeypatch.setenv("TC_LDSM_LATE_LAST", "0")
    cs0 = _check(spec, dcmotor_data(9), tape_oracle, SOLVE_ARGS, W=4, fuse=True)
    f0 = np.asarray(cs0["headers"], dtype=np.int64).reshape(-1, 8)
    assert not ((f0[(f0[:, 0] >> 8) == lsch.K_FOLD][:, 2] & lsch.LATE_LAST) != 0).any()


def test_byte_address_words_of_the_compiled_walk_decode_to_the_stream(native_built, tmp_path):
    """hip_emitter._ldsc_byte_words rewrites the body words of LDS-only two-phase records (a*2+sign -> a*8+sign, result
    and row addresses a -> a*8) and nothing else: decoding gives the stream back, word for word"""
    from tenscalc_amd import hip_emitter as he
    from tenscalc_amd.problems.mpc_dcmotor import mpc_dcmotor
    spec = mpc_dcmotor(folder=str(tmp_path))
    T = spec.tape.merged(lsch.fusion_sets(spec.tape))
    cs = lsch.class_stream(T, spec.classname, spec.defines, W=1)
    bw = np.asarray(he._ldsc_byte_words(cs), dtype=np.int64)
    w = np.asarray(cs["words"], dtype=np.int64)
    hd = np.asarray(cs["headers"], dtype=np.int64).reshape(-1, 8)
    changed = np.zeros(len(w), dtype=bool)
    nsplit = 0
    for r in hd:
        kind, n, off, fl, cn = int(r[0]) >> 8, int(r[1]), int(r[4]), int(r[5]), int(r[6])
        if kind not in (1, 3) or (fl & 3) or not he._ldsc_splittable(r):
            continue
        nsplit += 1
        nlm = (fl >> lsch.F_NLATE_SHIFT) & 3 if kind == 3 else 0
        na = 1 + n if kind == 1 else (12 if nlm else 5)
        S = -(-na // 4) * 4
        b, o = bw[off:off + S * cn].reshape(cn, S), w[off:off + S * cn].reshape(cn, S)
        ops = list(range(1, 1 + n)) if kind == 1 else [2, 3] + (list(range(6, 12)) if nlm else [])
        addr = [0] if kind == 1 else [0, 4]
        for c_ in ops:
            assert np.array_equal(((b[:, c_] >> 3) << 1) | (b[:, c_] & 1), o[:, c_]) and not (b[:, c_] & 6).any()
        for c_ in addr:
            assert np.array_equal(b[:, c_] >> 3, o[:, c_]) and not (b[:, c_] & 7).any()
        rest = [c_ for c_ in range(S) if c_ not in ops + addr]
        assert np.array_equal(b[:, rest], o[:, rest])
        changed[off:off + S * cn] = True
    assert nsplit > 100 and np.array_equal(bw[~changed], w[~changed])
    assert bw.max() < 2 ** 31


def test_class_digest_covers_exactly_the_files_a_class_includes():
    """hip_emitter._deps_of: the include closure of a class source -- an edit of csrc/tc_ldsm.hip.inc rebuilds the LDS
    machine's classes and nobody else's"""
    from tenscalc_amd import hip_emitter as he
    names = lambda src: {os.path.basename(p) for p in he._deps_of(src)}      # noqa: E731
    code = '#include "tc_wave.hip.inc"\n#include "tc_pow.hip.inc"\n#include "ipm_newton.hip.inc"\n#include "tc_runtime.hip.inc"\n'
    d = names(code)
    assert {"tc_wave.hip.inc", "tc_pow.hip.inc", "tc_glibcpow.hip.inc", "tc_glibcpow_tables.inc", "tc_crmath.hip.inc",
            "ipm_newton.hip.inc", "tc_runtime.hip.inc", "tenscalc_hip.h"} <= d
    assert "tc_ldsm.hip.inc" not in d and "tc_interp.hip.inc" not in d
    d = names('#include "tc_ldsm.hip.inc"\n')
    assert {"tc_ldsm.hip.inc", "tc_interp.hip.inc", "tc_pow.hip.inc"} <= d and "tc_wave.hip.inc" not in d


@pytest.mark.parametrize("W", [1, 4])
def test_ldsm_stream_groups_one_solve_cannot_tell_apart_are_merged(native_built, tape_oracle, tmp_path, W):
    """ldsm_schedule.loop_classes: dependency groups evaluated at the same call sites of the Newton loop and invalidated
    by the same entry points become one group (fewer hand-overs, joint level schedule); the sets are convex and leave the
    group graph acyclic; every get still reproduces the oracle, which walks the reference's partition"""
    from tenscalc_amd.problems.mpc_dcmotor import SOLVE_ARGS, dcmotor_data, mpc_dcmotor
    spec = mpc_dcmotor(folder=str(tmp_path))
    T1 = spec.tape.merged(lsch.fusion_sets(spec.tape))
    cl = lsch.loop_classes(T1, spec.defines)
    assert cl and all(len(S) > 1 for S in cl) and len({g for S in cl for g in S}) == sum(len(S) for S in cl)
    T2 = lsch.fused_tape(spec.tape, spec.defines)                       # (Tape.merged refuses cycles)
    assert T2.nGroups == T1.nGroups - sum(len(S) - 1 for S in cl) < T1.nGroups * 0.7
    assert max(len(m_) for m_ in T2.groups) >= max(len(m_) for m_ in T1.groups)      # the first step's merge survives the second
    _check(spec, dcmotor_data(11), tape_oracle, SOLVE_ARGS, W, fuse="loop")


def test_ldsm_stream_small_quadcopter_loop_merged(native_built, tape_oracle, tmp_path):
    from tenscalc_amd.problems.mpc_quadcopter import SOLVE_ARGS, mpc_quadcopter, quadcopter_data
    spec = mpc_quadcopter(T=8, classname="tmp_quadcopter8", folder=str(tmp_path))
    _check(spec, quadcopter_data(2, T=8), tape_oracle, SOLVE_ARGS, W=4, fuse="loop")


def test_split_div_tape_keeps_values_levels_and_groups(native_built, tape_oracle, tmp_path, monkeypatch):
    """ldsm_schedule.split_div_tape on a factorisation with long rows (mpc_quadcopter, T = 20): every split instruction is a
    division whose numerator is the same chain; the dependency groups and their level counts are those of the tape it
    was given; an oracle walking the REWRITTEN tape returns the outputs of the original bit for bit"""
    from tenscalc_amd.problems.mpc_quadcopter import SOLVE_ARGS, mpc_quadcopter, quadcopter_data
    spec = mpc_quadcopter(T=20, classname="tmp_quadcopter20", folder=str(tmp_path))
    T1 = lsch.fused_tape(spec.tape, spec.defines)
    T2 = lsch.split_div_tape(T1)
    nsplit = T2.n - T1.n
    assert nsplit > 50 and T2.nGroups == T1.nGroups
    # levels: never more than before (an instruction is split only where its divisor is what it waits for), and fewer in the
    # factorisation -- with the quotients spelled out, the terms built on them are evaluated from the numerators and the
    # pivot one level earlier (ldsm_schedule._div_nest): one fold step per column instead of two
    n1 = lsch.stream_levels(T1)[1]
    assert lsch.stream_levels(T2)[1] == n1                     # the split alone keeps every level
    monkeypatch.setenv("TC_LDSM_DIV_NEST", "1")
    n2 = lsch.stream_levels(T2)[1]
    assert all(b_ <= a_ for a_, b_ in zip(n1, n2)) and sum(n2) <= sum(n1) - 10
    monkeypatch.delenv("TC_LDSM_DIV_NEST")
    assert [len(m_) for m_ in T2.groups] != [len(m_) for m_ in T1.groups]           # the numerators joined their groups
    assert int((T2.types == I.I_div).sum()) - int((T1.types == I.I_div).sum()) == nsplit
    d = quadcopter_data(3, T=20)
    c = spec.classname
    outs = []
    for T in (spec.tape, T2):
        o = tape_oracle(T, c)
        for nm, _ in spec.parameters + spec.variables:
            o.set("%s_set_%s" % (c, nm), d[nm])
        st, it = o.solve(SOLVE_ARGS["mu0"], SOLVE_ARGS["maxIter"], -1, SOLVE_ARGS["addEye2Hessian"])
        outs.append((st, it, o.get(c + "_getOutputs")))
    assert outs[0][:2] == outs[1][:2] and outs[0][0] == 0
    assert np.array_equal(outs[0][2], outs[1][2])


def test_ldsm_stream_quadcopter20_div_nested_terms(native_built, tape_oracle, tmp_path, monkeypatch):
    """a factorisation with long rows (T = 20) on the tape the emitter walks: the stream holds late terms of the
    (N1/D) * ((N2/D) * D) form -- inline in fold records (aux bits DN_SHIFT..) -- and the model of the walk still reproduces
    every get of the oracle, which evaluates the ORIGINAL tape"""
    from tenscalc_amd.problems.mpc_quadcopter import SOLVE_ARGS, mpc_quadcopter, quadcopter_data
    monkeypatch.setenv("TC_LDSM_DIV_NEST", "1")          # (an option of the LDS machine, off by default: fewer steps, slower ones)
    spec = mpc_quadcopter(T=20, classname="tmp_quadcopter20", folder=str(tmp_path))
    cs = _check(spec, quadcopter_data(2, T=20), tape_oracle, SOLVE_ARGS, W=4, fuse="loop+split")
    hd = np.asarray(cs["headers"]).reshape(-1, 8)
    folds = hd[(hd[:, 0] >> 8) == lsch.K_FOLD]
    dots = folds[np.isin(folds[:, 0] & 255, list(lsch._DOT_SKIP))]
    assert ((dots[:, 2] >> lsch.DN_SHIFT) & 3).any(), "no fold record with a division-nested late term"


def test_ldsm_stream_dcmotor_as_benchmarked(native_built, tape_oracle, tmp_path):
    """the stream of bench.py's mpc_dcmotor class (`ldscs`: two wavefronts, a quarter of a CU's LDS, 512 staging slots, long
    reductions folded in two levels from 48 terms on -- the 96-term cost row as the reference's two statement lines)"""
    import __graft_entry__ as g
    from tenscalc_amd.problems.mpc_dcmotor import SOLVE_ARGS, dcmotor_data, mpc_dcmotor
    kw = g.VARIANT_KW["ldscs"]
    spec = mpc_dcmotor(folder=str(tmp_path))
    cs = _check(spec, dcmotor_data(9), tape_oracle, SOLVE_ARGS, W=kw["waves"], lds_slots=kw["lds_slots"], fuse="loop",
                stream_kw=dict(stage=kw["lds_stage"], split=kw["lds_split_min"]))
    hd = np.asarray(cs["headers"]).reshape(-1, 8)
    folds = hd[(hd[:, 0] >> 8) == lsch.K_FOLD]
    sp = folds[(folds[:, 0] & 255) == I.I_sumprod]
    assert len(sp) and sp[:, 1].max() <= 56 and cs["nCold"] == 0     # no sum of products longer than a statement line; state in LDS


def test_ldsm_stream_prefix_folds_option(native_built, tape_oracle, tmp_path, monkeypatch):
    """TC_LDSM_PREFIX=k (an option of the stream builder, off by default: measured slower, DESIGN 3.9 "round 6" h): the leading
    terms of a hoisted dot that were complete three levels up are staged two steps ahead and folded one step ahead into the
    first slot of the row -- the stream holds F_SUM folds whose destination is a staging slot, validate() finds no race among
    the four staging buffers, and the model of the walk still reproduces every get of the oracle"""
    from tenscalc_amd.problems.mpc_quadcopter import SOLVE_ARGS, mpc_quadcopter, quadcopter_data
    monkeypatch.setenv("TC_LDSM_PREFIX", "4")
    spec = mpc_quadcopter(T=20, classname="tmp_quadcopter20", folder=str(tmp_path))
    cs = _check(spec, quadcopter_data(4, T=20), tape_oracle, SOLVE_ARGS, W=4, fuse="loop", stream_kw=dict(stage=4096))
    hd = np.asarray(cs["headers"]).reshape(-1, 8)
    sums = hd[((hd[:, 0] >> 8) == lsch.K_FOLD) & ((hd[:, 0] & 255) == lsch.F_SUM)]
    lo, hi = cs["stage_base"], cs["stage_base"] + cs["nStage"]
    into_stage = 0
    for h in sums:
        dst = lsch.record_body(cs["words"], h)[0][:int(h[3])]
        into_stage += int(((dst >= lo) & (dst < hi)).sum())
    assert into_stage > 20, "no prefix fold writes into a staging row"


def test_ldsm_stream_merged_folds_option(native_built, tape_oracle, tmp_path, monkeypatch):
    """TC_LDSM_MERGE_FOLDS=1 (an option of the stream builder, off by default: no gain measured, DESIGN 3.9 "round 6" i): folds
    -(acc) of a hoisted level join the record of the level's folds h0 - (acc) as (-0.0) - (acc) -- fewer fold records, no
    I_minus_dot fold left beside an I_plus_minus_dot one in a step, and the model of the walk reproduces every get of the oracle"""
    from tenscalc_amd import instructions as I
    from tenscalc_amd.problems.mpc_quadcopter import SOLVE_ARGS, mpc_quadcopter, quadcopter_data
    spec = mpc_quadcopter(T=20, classname="tmp_quadcopter20", folder=str(tmp_path))
    T = lsch.fused_tape(spec.tape, spec.defines)

    def folds(cs):
        hd = np.asarray(cs["headers"]).reshape(-1, 8)
        return hd[(hd[:, 0] >> 8) == lsch.K_FOLD]
    n0 = len(folds(lsch.class_stream(T, spec.classname, spec.defines, W=4)))
    monkeypatch.setenv("TC_LDSM_MERGE_FOLDS", "1")
    cs = _check(spec, quadcopter_data(4, T=20), tape_oracle, SOLVE_ARGS, W=4, fuse="loop")
    f1 = folds(cs)
    assert len(f1) < n0
    assert (f1[:, 0] & 255 == I.I_plus_minus_dot).any()
