"""Tape -> HIP lowering (the step the build retargets: SURVEY.md 8a row A9).

The reference prints the scalar tape as straight-line C on one global scratchbook
(/root/reference/lib/@csparse/writeCfunctionpergroup.m:386-504 + ``writeCinstructionsC``,
lib/csparse/instructionsTableUTHash.c:790-1609).  This emitter prints the *same tape* as
CDNA4 HIP where **one thread owns one problem instance**:

* parameters and initial values live in HBM as structure-of-arrays ``S[element][B]`` so that
  a wavefront's 64 instances read 64 consecutive doubles (the batch is the coalesced axis);
* values that cross dependency-group boundaries live in a per-thread private array ``m[]``;
* values used only inside their group are SSA temporaries (registers);
* ``I_load`` constants are literals;
* each dependency group is one ``__device__`` function, evaluated lazily through a per-thread
  bitmask, exactly the ``groupStatus[]`` protocol of the reference
  (writeCfunctionpergroup.m:44,653-657);
* the Newton loop (csrc/ipm_newton.hip.inc) runs on the device, so a whole solve is one launch.

Per-opcode arithmetic (operand order, ``m[r]+=`` chunking of long sums) follows
instructionsTableUTHash.c:950-1599 so that with ``fp_contract='off'`` results are bit-identical
to the CPU oracle; ``fp_contract='fast'`` lets the compiler form FMAs.
"""
from __future__ import annotations

import hashlib
import os
import subprocess

import numpy as np

from . import instructions as I

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
MAX_TERMS_PERLINE = 100  # instructionsTableUTHash.c:173

_UNARY = {
    I.I_exp: "TC_EXP({0})", I.I_log: "TC_LOG({0})", I.I_cos: "TC_COS({0})", I.I_minus_cos: "-TC_COS({0})",
    I.I_sin: "TC_SIN({0})", I.I_minus_sin: "-TC_SIN({0})", I.I_round: "round({0})",
    I.I_ceil: "ceil({0})", I.I_floor: "floor({0})", I.I_abs: "fabs({0})",
    I.I_sign: "(({0}>=0)?1.0:-1.0)", I.I_sqrt: "sqrt({0})", I.I_Dsqrt: ".5*TC_POW_NEG({0},-.5)",
    I.I_DDsqrt: "-.25*TC_POW_NEG({0},-1.5)", I.I_sqr: "{0}*{0}", I.I_2times: "2*{0}", I.I_2: "2.0",
    I.I_cube: "{0}*{0}*{0}", I.I_3sqr: "3*{0}*{0}", I.I_6times: "6*{0}", I.I_atan: "TC_ATAN({0})",
    I.I_Datan: "1/(1+{0}*{0})", I.I_DDatan: "-2*{0}/(1+{0}*{0})/(1+{0}*{0})",
    I.I_srelu: "TC_LOG(1+TC_EXP({0}))", I.I_dsrelu: "1/(1+TC_EXP(-{0}))",
    I.I_ddsrelu: "1/(2+TC_EXP(-{0})+TC_EXP({0}))", I.I_relu: "(({0}>0)?{0}:0.0)",
    I.I_heaviside: "(({0}>0)?1.0:0.0)", I.I_inv: "TC_DIV(1.0,{0})", I.I_minus_inv_sqr: "-1/({0}*{0})",
    I.I_2_inv_cube: "2/({0}*{0}*{0})",
}


def _lit(v):
    v = float(v)
    if v != v or v in (float("inf"), float("-inf")):
        raise ValueError("non-finite constant in tape")
    s = repr(v)
    if "e" not in s and "." not in s:
        s += ".0"
    return "(" + s + ")" if v < 0 else s


def _solver_of(spec):
    """name of the Newton-loop template the class pastes behind its tape: ipmPD_CSsolver (cmex2optimizeCS.m:95,286) or
    ipmPDeq_CSsolver (cmex2equilibriumLatentCS.m:401,451)"""
    # (classes rebuilt from tape / .cg files carry no attribute: the equilibrium solvers' #define block has nZ,
    # cmex2equilibriumLatentCS.m:418, the optimisation solvers' has not)
    return getattr(spec, "solver", None) or ("ipmPDeq_CSsolver" if "nZ" in spec.defines else "ipmPD_CSsolver")


def _define_block(spec):
    """the solver configuration as the reference's #define block (cmex2optimizeCS.m:303-331,
    cmex2equilibriumLatentCS.m:417-448): whichever of the known names the class's defines hold"""
    w = ["// solver configuration (reference: cmex2optimizeCS.m:303-331 / cmex2equilibriumLatentCS.m:417-448)\n"]
    for k in ["nU", "nG", "nF", "nZ", "nD", "nX", "nNu", "scaleCost", "scaleInequalities", "scaleEqualities", "coupledAlphas",
              "setAddEye2Hessian", "adjustAddEye2Hessian", "useInertia", "delta", "skipAffine",
              "smallerNewtonMatrix", "useLDL", "useUmfpack", "allowSave"]:
        if k in spec.defines:
            w.append(f"#define {k} {int(spec.defines[k])}\n")
    for k in ["gradTolerance", "addEye2HessianUtolerance", "addEye2Hessian1tolerance", "equalTolerance", "desiredDualityGap",
              "alphaMin", "alphaMax", "muFactorAggressive", "muFactorConservative"]:
        if k in spec.defines:
            w.append(f"#define {k} {_lit(spec.defines[k])}\n")
    w.append(f"#define TC_SOLVER_SYMBOL {_solver_of(spec)}\n")
    if int(spec.defines.get("verboseLevel", 0)) >= 3:
        # solverVerboseLevel >= 3 (cmex2optimizeCS.m:331 passes it to the template): the per-iteration table of
        # ipmPD_CSsolver.c:248-345,775-863 is recorded for one instance (csrc/ipm_newton.hip.inc: TC_V3) and written by
        # tc_trace.  Lower levels print nothing per iteration and generate nothing here.
        w.append(f"#define verboseLevel {int(spec.defines['verboseLevel'])}\n#define TC_VERBOSE3 1\n#define TC_V3_MAXROWS 4096\n")
    return "".join(w)


from .lowering import Family, Leaf, Lowering, Node  # noqa: E402


class Printer:
    """Expression printer for one Lowering (semantics: instructionsTableUTHash.c:950-1599)."""

    _DEN_POS = {I.I_div: 1, I.I_minus_dot_div: 0, I.I_plus_minus_dot_div: 1, I.I_inv: 0}

    def __init__(self, L, roll=True, recip=False, fuse=False):
        self.L = L
        self.roll = roll
        self.recip = recip          # x/d -> x*(1/d) with 1/d shared inside a group (opt-in: changes rounding)
        self._rc = {}               # denominator instruction id -> name of its reciprocal in this group
        self.fuse = fuse            # long sums over the inputs share ONE sweep (see fused_sweeps)
        self._fused = {}            # sum instruction id -> name of its accumulated value in this group
        self._fused_d = {}          # distributed reduction id -> (name of its per-lane partial sum, scale)
        self._prefused = {}         # ... already accumulated by a sweep shared by several groups (kernel prologue)
        self._staged = {}           # exact lowering: distributed reduction id -> text of its (replicated) value
        # store->load forwarding inside a group (lanes>1): a value just written to LDS by this lane is
        # read back from the register that holds it -- a replicated value by every later statement of
        # the group, element c_ of a lane-split family by the aligned member c_ of a later family --
        # instead of through an LDS round trip the compiler cannot remove (it does not know that the
        # other lanes of the instance write other slots)
        self.forward = False
        self.sched_every = 0        # a scheduling barrier after every sched_every single operands of a lane reduction
        self.fw_window = 0          # forward a value only to the next fw_window units of its group (0: to all of them);
        #                             in a 647-instruction LDL every early factor entry would otherwise stay live in a
        #                             register to the end of the group -- 1 500 scratch accesses in mpc_dcmotor's
        self.branch_free_tails = True
        # exact=True (lanes>1): the wave-cooperative lowering keeps the reference's arithmetic bit for bit --
        # sums over lane-distributed operands are accumulated in the reference's term order (staged through LDS:
        # exact_dsweeps), sums over replicated operands are not lane-split, min/max/clp butterflies fall back to
        # the reference's sequential chain when the result is NaN or a zero (whose sign is order dependent)
        self.exact = False
        self.stg_base = self.res_base = 0      # first LDS slot of the staging buffers / of the staged results
        self.stg_blocks = 1                    # row blocks (of P terms) staged per LDS round trip
        self.ilp = False                       # print independent chains interleaved (see dfamily_body)
        # staged terms loaded back to back before their additions: 16 measured best in round 2; with the y family out of
        # the loop and the ILP scheduling strategy 8 leaves no register spilled (16: 7, 32: 33) and is 1 % faster
        self.stg_batch = int(os.environ.get("TC_STG_BATCH", "8"))
        self._rv = {}               # slot -> index into c.rv[]: register copy that outlives the group (setup_rcache)
        self._fw_rep = {}           # slot -> name of the register copy (replicated values)
        self._fw_fam = {}           # (first slot, slot stride) -> (name prefix of per-block copies, members)

    def _den_leaves(self, t, acc):
        """denominator leaves of every division in tree t"""
        p = self._DEN_POS.get(t.typ)
        if p is not None and isinstance(t.kids[p], Leaf) and t.kids[p].kind in ("priv", "tmp"):
            acc.append(t.kids[p])
        for k in t.kids:
            if isinstance(k, Node):
                self._den_leaves(k, acc)
        return acc

    # leaf text ---------------------------------------------------------------------------------
    @staticmethod
    def leaf_text(lf, base=None, dc=0, dk=0, var="c_"):
        if lf.kind == "lit":
            return _lit(lf.val)
        if lf.kind == "tmp":
            return f"t{lf.idx}"
        b = lf.idx if base is None else base
        ix = str(b)
        if dc:
            ix += f"+({dc})*{var}" if dc != 1 else f"+{var}"
        if dk:
            ix += f"+{dk}*k_" if dk != 1 else "+k_"
        return f"SX({ix})" if lf.kind == "ext" else f"MM({ix})"

    def vec_ref(self, vec, k):
        """element block k of a distributed vector as seen by one lane (register array)"""
        L = self.L
        if vec[0] == "X":
            d = [x for x in L.inputs if x["dist"] and x["arr"] == vec[1]][0]
            if not d.get("resident", True):
                # too large to keep in registers: read this lane's element from the distributed
                # block on every use (still one 512-byte line per wave-load)
                # uniform base + 32-bit byte offset (global_load with scalar base): a 64-bit per-lane address
                # for each of the ~120 elements would otherwise sit in registers for the whole kernel
                if isinstance(k, str):          # rolled sweep: row block index is a loop variable
                    return f"TC_SD({d['off'] + vec[2] * d['K']} + {k})"
                return f"TC_SD({d['off'] + vec[2] * d['K'] + k})"
            return f"c.dX{vec[1]}[{vec[2] * d['K'] + k}]"
        return f"c.dF{vec[1]}[{k}]"

    def _reduction_terms(self, t, symbolic=False, guard=True):
        """per-lane terms (one per block of P rows) of a sum over aligned distributed operands, and the
        power-of-two literal factor taken out of the sum"""
        L = self.L
        nf, ns, facs = L.dreduce[t.ins]
        P_ = L.P
        K = -(-ns // P_)
        # a literal factor that is a power of two scales every term exactly: take it out of the sum
        # (bit-identical result, one multiplication instead of one per term)
        import math
        scale = 1.0
        skip = set()
        for p, (kind, v) in enumerate(facs):
            lf = t.kids[p]
            if self.exact:
                break       # (2*a)*b and 2*(a*b) differ once a*b is subnormal: keep the reference's factors
            if kind == "r" and lf.kind == "lit" and lf.val != 0 and math.frexp(abs(lf.val))[0] == 0.5:
                scale *= lf.val
                skip.add(p)
        if len(skip) == len(facs):
            skip, scale = set(), 1.0
        terms = []
        for k in list(range(K)) + (["k_"] if symbolic else []):
            fs = []
            for p, (kind, v) in enumerate(facs):
                if p in skip:
                    continue
                if kind == "d":
                    fs.append(self.vec_ref(v, k))
                else:
                    fs.append(self.leaf_text(t.kids[p]))
            tk = f"{fs[0]}*{fs[0]}" if t.typ == I.I_plus_sqr else "*".join(fs)
            if guard and not isinstance(k, str) and (k + 1) * P_ > ns:
                tk = f"(({k * P_}+c.lane<{ns})?({tk}):0.0)"
            terms.append(tk)
        if symbolic:
            return terms[:-1], scale, terms[-1]
        return terms, scale

    def reduction(self, t):
        """sum over aligned distributed operands: per-lane partial sum + butterfly all-reduce"""
        if t.ins in self._staged:
            return self._staged[t.ins]
        if self.exact:
            raise AssertionError("exact lowering: distributed reduction outside a staged sweep")
        if t.ins in self._fused_d:
            e, scale = self._fused_d[t.ins]
            return f"tc_allreduce({e})" if scale == 1.0 else f"({_lit(scale)}*tc_allreduce({e}))"
        terms, scale = self._reduction_terms(t)
        e = terms[0]
        for tk in terms[1:]:
            e = f"({e})+{tk}"
        return f"tc_allreduce({e})" if scale == 1.0 else f"({_lit(scale)}*tc_allreduce({e}))"

    def _dreduce_nodes(self, t, acc):
        if isinstance(t, Leaf):
            return acc
        if t.ins in self.L.dreduce:
            acc.append(t)
            return acc
        for k in t.kids:
            self._dreduce_nodes(k, acc)
        return acc

    def fused_dsweeps(self, g):
        """lanes>1, distributed inputs NOT resident in registers (reg_budget=0: the hoisted tape needs them
        only before and after the Newton loop): all reductions of the group over those inputs (36+8 for
        hoisted sls) accumulate in ONE sweep over the row blocks, so a row block is loaded once (9 loads,
        44 multiply-adds) instead of being held in ~230 registers by 44 separate expressions.  Per-lane
        term order is unchanged."""
        L = self.L
        self._fused_d = dict(self._prefused)
        if L.P == 1 or not self.fuse:
            return ""
        nodes = []
        for g_ in (g if isinstance(g, (list, tuple)) else [g]):
            for u in L.units[g_ - 1]:
                if isinstance(u, (int, np.integer)):
                    self._dreduce_nodes(L.tree(int(u)), nodes)
        nodes = [t for t in nodes if t.ins not in self._prefused]

        def streamed(t):
            return all(kind != "d" or (v[0] == "X" and not [x for x in L.inputs if x["dist"] and x["arr"] == v[1]][0].get("resident", True))
                       for kind, v in L.dreduce[t.ins][2])
        nodes = [t for t in nodes if streamed(t)]
        if len(nodes) < 2:
            return ""
        per = [self._reduction_terms(t, symbolic=True) for t in nodes]
        K = max(len(terms) for terms, _, _ in per)
        if any(len(terms) != K for terms, _, _ in per):
            return ""
        import re
        ref_re = re.compile(r"TC_SD\([^()]*\)")

        def block(texts, tag):
            """one row block: every distinct input element is loaded once into a local"""
            loads, lines = {}, []
            for t, tk in zip(nodes, texts):
                for ref in sorted(set(ref_re.findall(tk))):
                    nm = loads.setdefault(ref, f"ld{tag}_{len(loads)}")
                    tk = tk.replace(ref, nm)
                lines.append(f"    fd{t.ins} = (fd{t.ins})+{tk};\n")
            return "".join(f"    const double {nm} = {ref};\n" for ref, nm in loads.items()) + "".join(lines)
        ragged = any("c.lane<" in terms[-1] for terms, _, _ in per)
        nroll = K - 1 if ragged else K
        w = "  double " + ", ".join(f"fd{t.ins} = 0.0" for t in nodes) + ";\n"
        # a real loop (not unrolled): otherwise the scheduler hoists every load of the sweep to the top and
        # holds all of the inputs in registers again
        un = os.environ.get("TC_SWEEP_UNROLL", "4")     # row blocks in flight per trip (loads overlap inside a trip)
        w += f"  _Pragma(\"unroll {un}\") for (int k_ = 0; k_ < {nroll}; k_++) {{\n" + block([sym for _, _, sym in per], "") + "  }\n"
        if ragged:
            w += "  {\n" + block([terms[-1] for terms, _, _ in per], "t") + "  }\n"
        per = [(terms, scale) for terms, scale, _ in per]
        for t, (terms, scale) in zip(nodes, per):
            self._fused_d[t.ins] = (f"fd{t.ins}", scale)
        return w

    def exact_dsweeps(self, g):
        """exact lowering (lanes>1): every sum of group g over lane-distributed operands, accumulated in the
        REFERENCE's term order (instructionsTableUTHash.c:987-1061: terms left to right, a new "m[r]+=" chain every
        MAX_TERMS_PERLINE factors), so the value is the one a single thread would compute, bit for bit.

        A butterfly all-reduce reassociates; a running sum handed from lane to lane would leave 7 of the 8 lanes
        idle (registers cannot be indexed per lane, so the lanes cannot work on different columns of a skewed
        pipeline).  Instead the sums are TRANSPOSED through LDS: the reductions of one shape are the columns, the
        terms the rows.  Per block of P rows every lane computes the products of ITS row for P columns and writes
        them to a staging buffer (P x P doubles, row stride P+1: conflict-free both ways); then lane c adds the P
        staged terms of column c to its running chain, in order.  All P lanes stay busy (P columns in flight), each
        product is computed once, each sum is a sequential chain in one lane.  Several row blocks are staged per
        round (as many as the LDS budget allows) so that a round costs one LDS round trip.  The totals end up in the staging area
        (dead by then), one slot per reduction, where the group's statements read them as replicated values."""
        L = self.L
        self._staged = {}
        if not self.exact or L.P == 1:
            return ""
        nodes = []
        for g_ in (g if isinstance(g, (list, tuple)) else [g]):
            for u in L.units[g_ - 1]:
                if isinstance(u, (int, np.integer)):
                    self._dreduce_nodes(L.tree(int(u)), nodes)
        if not nodes:
            return ""
        P_ = L.P
        RS = P_ + 1                                  # row stride of a staged column
        NB = self.stg_blocks                         # row blocks staged per round
        byshape = {}
        for t in nodes:
            nf, ns, facs = L.dreduce[t.ins]
            byshape.setdefault((t.typ, nf, ns), []).append(t)
        if sum(-(-len(v) // P_) * P_ for v in byshape.values()) > NB * P_ * RS:
            raise NotImplementedError("exact lowering: more distributed reductions in one group than staging slots")
        w = ["  { // sums over lane-distributed operands in the reference's term order, transposed through LDS\n"]
        res = 0
        tails = []
        for sh, (key, lst) in enumerate(byshape.items()):
            typ, nf, ns = key
            chunks = self._chunks(typ, nf, ns)
            start_of = {a: ci for ci, (a, _) in enumerate(chunks)}
            K = -(-ns // P_)
            R = len(lst)
            per = [self._reduction_terms(t, guard=False)[0] for t in lst]
            nq = -(-R // P_)
            for q in range(nq):
                nm = f"s{sh}_{q}"
                w.append(f"    double {nm}l_ = 0.0, {nm}t_ = 0.0;\n")
                ncol = min(P_, R - q * P_)
                for k0 in range(0, K, NB):
                    blocks = list(range(k0, min(K, k0 + NB)))
                    # every lane: the products of its row of each block, for the columns of this column block
                    for bi, k in enumerate(blocks):
                        for cc in range(ncol):
                            w.append(f"    MM({self.stg_base + bi * P_ * RS + cc * RS}+lane) = {per[q * P_ + cc][k]};\n")
                    w.append("    __builtin_amdgcn_wave_barrier();\n")
                    # lane c: the staged terms of column c, all loads first (one LDS round trip per round), then
                    # the chain of additions in row order
                    rows = [(bi, r) for bi, k in enumerate(blocks) for r in range(min(P_, ns - k * P_))]
                    for r0 in range(0, len(rows), self.stg_batch):
                        part = rows[r0:r0 + self.stg_batch]
                        w.append("    { const double " + ", ".join(
                            f"v{bi}_{r} = MM({self.stg_base + bi * P_ * RS + r}+lane*{RS})" for bi, r in part) + ";\n")
                        # keep the loads together: left alone, the scheduler (minimising register pressure) pairs
                        # every load with its addition and waits for each one in turn
                        w.append("      __builtin_amdgcn_sched_barrier(0);\n")
                        for bi, r in part:
                            row = blocks[bi] * P_ + r
                            v = f"v{bi}_{r}"
                            ci = start_of.get(row)
                            if ci is None:
                                w.append(f"      {nm}l_ = {nm}l_ + {v};\n")
                            else:
                                if ci == 1:
                                    w.append(f"      {nm}t_ = {nm}l_;\n")
                                elif ci > 1:
                                    w.append(f"      {nm}t_ = {nm}t_ + {nm}l_;\n")
                                w.append(f"      {nm}l_ = {v};\n")
                        w.append("    }\n")
                    w.append("    __builtin_amdgcn_wave_barrier();\n")
                tot = f"{nm}l_" if len(chunks) == 1 else f"({nm}t_ + {nm}l_)"
                guard = "" if ncol == P_ else f"if (lane < {ncol}) "
                tails.append(f"    {guard}MM({self.stg_base + res}+lane) = {tot};\n")
                for cc in range(ncol):
                    self._staged[lst[q * P_ + cc].ins] = f"MM({self.stg_base + res + cc})"
                res += P_
        w += tails
        w.append("    __builtin_amdgcn_wave_barrier();\n  }\n")
        return "".join(w)

    def dfamily_body(self, f):
        L = self.L
        K = -(-f.n // L.P)
        out = []
        t = f.template
        if self.ilp and t.typ == I.I_sumprod and K > 1 and int(t.par[0]) * int(t.par[1]) < MAX_TERMS_PERLINE:
            # the K rows of this lane are independent chains of multiply, add, multiply, add ...: printed row by
            # row every instruction waits for the one before it (a dependent fp64 instruction issues after 8
            # cycles, an independent one after 4: scripts/ubench/fp64_latency.hip).  Printed term by term ACROSS
            # the rows -- same operations, same order within a row, same values -- the chains overlap.
            nf, ns = int(t.par[0]), int(t.par[1])
            rows = []
            for k in range(K):
                pos = [0]
                r = []
                for kid in t.kids:
                    if isinstance(kid, Leaf):
                        kind, v = f.leaves[pos[0]]
                        pos[0] += 1
                        r.append(self.vec_ref(v, k) if kind == "d" else self.leaf_text(v))
                    else:
                        r.append("(" + self._dexpr(kid, f, k, pos) + ")")
                rows.append(["*".join(r[j * nf:(j + 1) * nf]) for j in range(ns)])
            out.append("  {\n    double " + ", ".join(f"q{k}_ = {rows[k][0]}" for k in range(K)) + ";\n")
            for j in range(1, ns):
                out.append("    { const double " + ", ".join(f"p{k}_ = {rows[k][j]}" for k in range(K)) + ";\n     "
                           + "".join(f" q{k}_ = q{k}_ + p{k}_;" for k in range(K)) + " }\n")
            out.append("   " + "".join(f" c.dF{f.fid}[{k}] = q{k}_;" for k in range(K)) + "\n  }\n")
            return "".join(out)
        for k in range(K):
            pos = [0]
            e = self._dexpr(f.template, f, k, pos)
            out.append(f"  c.dF{f.fid}[{k}] = {e};\n")
        return "".join(out)

    def _dexpr(self, t, f, k, pos):
        r = []
        for kid in t.kids:
            if isinstance(kid, Leaf):
                kind, v = f.leaves[pos[0]]
                pos[0] += 1
                r.append(self.vec_ref(v, k) if kind == "d" else self.leaf_text(v))
            else:
                r.append("(" + self._dexpr(kid, f, k, pos) + ")")
        return self.op_text(t.typ, t.par, r)

    # reductions over REPLICATED operands held in consecutive LDS slots: the P lanes of the instance
    # each take every P-th operand and combine with a DPP butterfly (sum / min / max)
    _LANE_RED = {I.I_min: ("min", 1), I.I_min0: ("min", 1), I.I_max: ("max", 1), I.I_max0: ("max", 1),
                 I.I_max_abs: ("max", 1), I.I_clp: ("min", 2), I.I_plus_sqr: ("add", 1), I.I_plus_abs: ("add", 1),
                 I.I_sumprod: ("add", 0), I.I_sum: ("add", 1)}

    def _col_text(self, kids):
        """text of one operand column over a run of terms (index variable j_), or None if the run
        is not affine"""
        L = self.L
        k0 = kids[0]
        n = len(kids)
        if isinstance(k0, Node):
            if not all(isinstance(x, Node) for x in kids):
                return None
            bd = L.affine_trees(kids)
            if bd is None:
                return None
            return "(" + self.expr(k0, Family([], k0, bd[0], bd[1], 0, 0, var="j_"), [0]) + ")"
        if not all(isinstance(x, Leaf) and x.kind == k0.kind for x in kids):
            return None
        if k0.kind == "lit":
            return _lit(k0.val) if all(x.val == k0.val for x in kids) else None
        if k0.kind == "tmp":
            return f"t{k0.idx}" if all(x.idx == k0.idx for x in kids) else None
        if k0.kind != "priv":
            return None
        d = kids[1].idx - k0.idx if n > 1 else 0
        if any(x.idx != k0.idx + d * j for j, x in enumerate(kids)):
            return None
        return f"MM({k0.idx}+({d})*j_)" if d else f"MM({k0.idx})"

    def lane_reduction(self, t):
        """min/max/clp/sum over many operands held in LDS: split the operand list into maximal
        affine runs, let the P lanes of the instance take every P-th operand of each run, combine
        with one DPP butterfly; operands outside any run are folded in afterwards."""
        L = self.L
        if L.P <= 1 or t.typ not in self._LANE_RED:
            return None
        kind, w = self._LANE_RED[t.typ]
        if self.exact and (kind == "add" or t.typ in (I.I_min0, I.I_max0)):
            return None         # a lane-split sum reassociates; min0/max0 start from a zero (see below)
        if t.typ == I.I_sumprod:
            w = int(t.par[0])
        n = len(t.kids) // w
        if n < L.P or len(t.kids) != n * w:
            return None
        if t.typ == I.I_sum and any(p != t.par[0] for p in t.par):
            return None
        ty = t.typ

        def step(cols):
            if ty in (I.I_min, I.I_min0):
                return f"a_ = tc_op_min(a_, {cols[0]});"
            if ty in (I.I_max, I.I_max0):
                return f"a_ = tc_op_max(a_, {cols[0]});"
            if ty == I.I_max_abs:
                return f"a_ = tc_op_max(a_, fabs({cols[0]}));"
            if ty == I.I_clp:
                return f"a_ = tc_clp(a_, {cols[0]}, {cols[1]});"
            if ty == I.I_plus_sqr:
                return f"a_ = a_ + {cols[0]}*{cols[0]};"
            if ty == I.I_plus_abs:
                return f"a_ = a_ + fabs({cols[0]});"
            if ty == I.I_sum:
                return f"a_ = a_ + {cols[0]};"
            return "a_ = a_ + " + "*".join(cols) + ";"

        minrun = max(2, L.P // 2)
        runs, singles = [], []
        k0 = 0
        while k0 < n:
            best = None
            k1 = min(n, k0 + minrun)
            while k1 <= n:
                cols = [self._col_text([t.kids[k * w + p] for k in range(k0, k1)]) for p in range(w)]
                if any(c_ is None for c_ in cols) or (k1 - k0 > 1 and not any("j_" in c_ for c_ in cols)):
                    break
                best = (k1, cols)
                k1 += 1
            if best is None:
                singles.append(k0)
                k0 += 1
            else:
                runs.append((k0, best[0], best[1]))
                k0 = best[0]
        if not runs or sum(k1 - k0 for k0, k1, _ in runs) < L.P:
            return None
        init = {I.I_min: "DBL_MAX", I.I_max: "-DBL_MAX", I.I_clp: "DBL_MAX"}.get(ty, "0.0")
        if self.exact:
            init = {I.I_min: "__builtin_inf()", I.I_max: "(-__builtin_inf())", I.I_clp: "DBL_MAX"}.get(ty, "0.0")
        red = {"add": "tc_allreduce", "min": "tc_allreduce_min", "max": "tc_allreduce_max"}[kind]
        body = f"({{ double a_ = {init};"
        for k0, k1, cols in runs:
            for q in range(-(-(k1 - k0) // L.P)):
                guard = "" if (q + 1) * L.P <= k1 - k0 else f"if (j_ < {k1 - k0}) "
                body += f" {{ const int j_ = lane + {q * L.P}; {guard}{{ {step(cols)} }} }}"
        body += f" a_ = {red}(a_);"
        for q, k in enumerate(singles):
            cols = []
            for p in range(w):
                kid = t.kids[k * w + p]
                cols.append(self.leaf_text(kid) if isinstance(kid, Leaf) else "(" + self.expr(kid) + ")")
            body += " " + step(cols)
            if self.sched_every and len(singles) > 2 * self.sched_every and q % self.sched_every == self.sched_every - 1:
                # hundreds of operands outside any affine run (the rows of WW*dx - b in mpc_dcmotor's direction
                # error): left alone the scheduler requests the LDS operands of ALL of them up front and spills
                # (1 500 scratch accesses in that one group)
                body += " __builtin_amdgcn_sched_barrier(0);"
        if self.exact:
            # min/max of the same operands is order independent EXCEPT where the reference's chain
            # "if (acc > x) acc = x" depends on position: a NaN operand (kept only in first position, skipped
            # elsewhere) and equal zeros of opposite sign (the first one stays).  The butterfly propagates NaN, so
            # both cases show in the result and take the reference's own sequential chain (cold path; every lane
            # of the instance holds the same a_, so the instance's lanes branch together)
            r = []
            for kid in t.kids:
                r.append(self.leaf_text(kid) if isinstance(kid, Leaf) else "(" + self.expr(kid) + ")")
            body += " if (__builtin_expect(a_ != a_ || a_ == 0.0, 0)) { a_ = " + self.op_text(ty, t.par, r) + "; }"
        body += " a_; })"
        if ty == I.I_sum and t.par[0] < 0:
            body = "-" + body
        return body

    def expr(self, t, fam=None, pos=None):
        """C expression of tree t.  fam: Family (leaf index = base+delta*c_) or None.
        pos: running index into the flattened leaf list (list of one int)."""
        if pos is None:
            pos = [0]
        L = self.L
        if t.ins in L.dreduce:
            pos[0] += len(t.kids)
            return self.reduction(t)
        if fam is None and L.P > 1:
            lr = self.lane_reduction(t)
            if lr is not None:
                pos[0] += len(t.kids)
                return lr
        base = fam.base if fam else None
        delta = fam.delta if fam else None
        # V2: long sums rolled over the term index k_
        if (self.roll or self.fuse) and t.typ in (I.I_sumprod, I.I_plus_sqr) and all(isinstance(k, Leaf) for k in t.kids):
            lb = base if fam else [k.idx for k in t.kids]
            ld = delta if fam else [0] * len(t.kids)
            p0 = pos[0] if fam else 0
            tf = Lowering.term_family(t, lb, ld, p0)
            if tf is not None and not self.roll and any(f[0] not in ("ext", "lit") for f in tf[2]):
                tf = None       # roll=False: only sums over the inputs stay loops (no indexed access to m[])
            if tf is not None:
                pos[0] += len(t.kids)
                return self._term_loop(t, tf)
        r = []
        for k in t.kids:
            if isinstance(k, Leaf):
                j = pos[0]
                pos[0] += 1
                fw = None
                if self.forward and k.kind == "priv":
                    b_, d_ = (base[j], delta[j]) if fam else (k.idx, 0)
                    if d_ == 0:
                        fw = self._fw_rep.get(b_)
                        if fw is None and b_ in self._rv:
                            fw = f"c.rv[{self._rv[b_]}]"
                    elif fam is not None and fam.var == "c_" and (b_, d_) in self._fw_fam \
                            and fam.n <= self._fw_fam[(b_, d_)][1]:
                        fw = "@FW" + self._fw_fam[(b_, d_)][0] + "@"          # per-block name, see group_body
                r.append(fw or self.leaf_text(k, base[j] if fam else None, delta[j] if fam else 0,
                                              var=fam.var if fam else "c_"))
            else:
                r.append("(" + self.expr(k, fam, pos) + ")")
        p = self._DEN_POS.get(t.typ)
        if fam is None and p is not None and isinstance(t.kids[p], Leaf) and t.kids[p].ins in self._rc:
            rc = self._rc[t.kids[p].ins]
            if t.typ == I.I_inv:
                return rc
            r2 = list(r)
            r2[p] = "1.0"
            return "(" + self.op_text(t.typ, t.par, r2).replace("TC_DIV", "TC_NUM") + ")*" + rc
        return self.op_text(t.typ, t.par, r)

    # ---- fused sweeps -----------------------------------------------------------------------
    @staticmethod
    def _chunks(typ, nf, ns):
        """[(first term, end term)] of the reference's "m[r]+=" statements (:1003-1012)"""
        bounds, nSums, k0 = [], 0, 0
        if typ == I.I_sumprod:
            for k in range(ns):
                if k + 1 < ns:
                    nSums += nf
                    if nSums >= MAX_TERMS_PERLINE:
                        bounds.append((k0, k + 1))
                        k0, nSums = k + 1, 0
        bounds.append((k0, ns))
        return bounds

    def _fusable(self, t, acc):
        if isinstance(t, Leaf):
            return acc
        if t.typ in (I.I_sumprod, I.I_plus_sqr) and all(isinstance(k, Leaf) for k in t.kids):
            tf = Lowering.term_family(t, [k.idx for k in t.kids], [0] * len(t.kids), 0)
            if tf is not None and all(f[0] in ("ext", "lit") for f in tf[2]) and any(f[0] == "ext" for f in tf[2]):
                acc.append((t, tf))
            return acc
        for k in t.kids:
            self._fusable(k, acc)
        return acc

    def fused_sweeps(self, g):
        """One thread per instance: every long sum of group g whose terms read only the inputs
        (A'A, A'b, ...) is accumulated in ONE sweep over the term index, so each input element is
        loaded once per sweep instead of once per sum (36+8 sums over A for hoisted sls: one pass over
        A instead of nine).  Term order and the reference's chunking of each sum are kept, so the
        values are those of the separate loops, bit for bit."""
        L = self.L
        self._fused = {}
        if not (self.fuse and L.P == 1):
            return ""
        cands = []
        for u in L.units[g - 1]:
            if isinstance(u, (Family,)) or not isinstance(u, (int, np.integer)):
                continue
            self._fusable(L.tree(int(u)), cands)
        byn = {}
        for t, tf in cands:
            byn.setdefault(tf[1], []).append((t, tf))
        out = []
        for ns, lst in byn.items():
            if len(lst) < 2:
                continue
            cuts = {0, ns}
            info = []
            for t, (nf, _, facs) in lst:
                ch = self._chunks(t.typ, nf, ns)
                info.append((t, nf, facs, ch))
                for a, b in ch:
                    cuts.update((a, b))
                self._fused[t.ins] = f"fs{t.ins}"
            cuts = sorted(cuts)

            def term(t, facs, kexpr):
                fs = []
                for kind, b0, dk, dc, l0 in facs:
                    if kind == "lit":
                        fs.append(_lit(l0.val))
                    else:
                        fs.append(f"SX({b0}+{dk}*({kexpr}))" if dk else f"SX({b0})")
                return f"{fs[0]}*{fs[0]}" if t.typ == I.I_plus_sqr else "*".join(fs)
            w = "  double " + ", ".join(f"fs{t.ins}, fl{t.ins}" for t, *_ in info) + ";\n"
            for a, b in zip(cuts[:-1], cuts[1:]):
                w += "  {\n"
                for t, nf, facs, ch in info:
                    op = "=" if any(a == c0 for c0, _ in ch) else f"= fl{t.ins} +"
                    w += f"    fl{t.ins} {op} {term(t, facs, str(a))};\n"
                if b - a > 1:
                    w += f"    _Pragma(\"unroll 2\") for (int k_ = {a + 1}; k_ < {b}; k_++) {{\n"
                    for t, nf, facs, ch in info:
                        w += f"      fl{t.ins} = fl{t.ins} + {term(t, facs, 'k_')};\n"
                    w += "    }\n"
                for t, nf, facs, ch in info:
                    for q, (c0, c1) in enumerate(ch):
                        if c1 == b:
                            w += f"    fs{t.ins} = fl{t.ins};\n" if q == 0 else f"    fs{t.ins} = fs{t.ins} + fl{t.ins};\n"
                w += "  }\n"
            out.append(w)
        return "".join(out)

    def _term_loop(self, t, tf):
        if t.ins in self._fused:
            return self._fused[t.ins]
        nf, ns, facs = tf

        def term(kexpr):
            fs = []
            for kind, b0, dk, dc, l0 in facs:
                if kind == "lit":
                    fs.append(_lit(l0.val))
                elif kind == "tmp":
                    fs.append(f"t{l0.idx}")
                else:
                    ix = f"{b0}"
                    if dc:
                        ix += f"+{dc}*c_" if dc != 1 else "+c_"
                    if dk:
                        ix += f"+{dk}*({kexpr})" if dk != 1 else f"+({kexpr})"
                    fs.append(f"SX({ix})" if kind == "ext" else f"MM({ix})")
            if t.typ == I.I_plus_sqr:
                return f"{fs[0]}*{fs[0]}"
            return "*".join(fs)
        # chunk boundaries: a new "m[r]+=" statement every MAX_TERMS_PERLINE factors (:1003-1012)
        bounds, nSums, k0 = [], 0, 0
        if t.typ == I.I_sumprod:
            for k in range(ns):
                if k + 1 < ns:
                    nSums += nf
                    if nSums >= MAX_TERMS_PERLINE:
                        bounds.append((k0, k + 1))
                        k0, nSums = k + 1, 0
        bounds.append((k0, ns))
        s = "({ double tot_ = 0, ln_;"
        for q, (a, b) in enumerate(bounds):
            s += f" ln_ = {term(str(a))};"
            if b - a > 1:
                s += f" _Pragma(\"unroll 4\") for (int k_ = {a + 1}; k_ < {b}; k_++) ln_ = ln_ + {term('k_')};"
            s += " tot_ = ln_;" if q == 0 else " tot_ = tot_ + ln_;"
        return s + " tot_; })"

    @staticmethod
    def op_text(typ, par, r):
        if typ == I.I_sum:
            s = ("+" if par[0] > 0 else "-") + r[0]
            for p, x in zip(par[1:], r[1:]):
                s += ("+" if p > 0 else "-") + x
            return s
        if typ == I.I_sumprod:
            nf, ns = int(par[0]), int(par[1])
            terms = ["*".join(r[k * nf:(k + 1) * nf]) for k in range(ns)]
            lines, cur, nSums = [], [], 0
            for k, t in enumerate(terms):
                cur.append(t)
                if k + 1 < ns:
                    nSums += nf
                    if nSums >= MAX_TERMS_PERLINE:
                        lines.append(cur)
                        cur, nSums = [], 0
            lines.append(cur)
            e = "+".join(lines[0])
            for ln in lines[1:]:
                e = f"({e})+({'+'.join(ln)})"
            return e
        if typ == I.I_div:
            return f"TC_DIV({r[0]},{r[1]})"
        if typ in (I.I_minus_dot, I.I_minus_dot_div, I.I_plus_minus_dot, I.I_plus_minus_dot_div):
            skip = {I.I_minus_dot: 0, I.I_minus_dot_div: 1, I.I_plus_minus_dot: 1,
                    I.I_plus_minus_dot_div: 2}[typ]
            d = r[skip:]
            dot = "+".join(f"{d[k]}*{d[k + 1]}" for k in range(0, len(d), 2))
            if typ == I.I_minus_dot:
                return f"-({dot})"
            if typ == I.I_minus_dot_div:
                return f"TC_DIV(-({dot}),{r[0]})"
            if typ == I.I_plus_minus_dot:
                return f"{r[0]}-({dot})"
            return f"TC_DIV(({r[0]}-({dot})),{r[1]})"
        if typ == I.I_plus_sqr:
            return "+".join(f"{x}*{x}" for x in r)
        if typ == I.I_plus_abs:
            return "+".join(f"fabs({x})" for x in r)
        if typ in (I.I_min, I.I_max, I.I_min0, I.I_max0, I.I_max_abs, I.I_clp):
            # acc = first (or 0 / DBL_MAX); then one conditional update per operand, in operand order
            # (instructionsTableUTHash.c:1227-1331).  Long lists are printed as a statement sequence:
            # nesting hundreds of calls overflows the compiler's recursive descent.
            f = "tc_min" if typ in (I.I_min, I.I_min0) else "tc_max"
            if typ in (I.I_min, I.I_max):
                init, steps = r[0], [f"{f}(a_,{x})" for x in r[1:]]
            elif typ in (I.I_min0, I.I_max0):
                init, steps = "0.0", [f"{f}(a_,{x})" for x in r]
            elif typ == I.I_max_abs:
                init, steps = f"fabs({r[0]})", [f"tc_max(a_,fabs({x}))" for x in r[1:]]
            else:
                init, steps = "DBL_MAX", [f"tc_clp(a_,{r[k]},{r[k + 1]})" for k in range(0, len(r), 2)]
            if len(steps) <= 12:
                e = init
                for st_ in steps:
                    e = st_.replace("a_", e, 1)
                return e
            return "({ double a_ = " + init + "; " + " ".join(f"a_ = {st_};" for st_ in steps) + " a_; })"
        if typ == I.I_zero:
            return "0.0"
        if typ == I.I_componentwise:      # the reference prints the C text verbatim with m[operand] in the hole (:1333-1345)
            eq = I.componentwise_opcode(par)      # a plain call of a function the ISA has an opcode for: that opcode's code,
            if eq in _UNARY:                      # so that exp / log / pow are the HOST libm's here too (csrc/tc_pow.hip.inc)
                return "(" + _UNARY[eq].format("(" + r[0] + ")") + ")"
            return "(" + I.componentwise_text(par).replace("%s", "(" + r[0] + ")") + ")"
        if typ in _UNARY:
            return _UNARY[typ].format(r[0] if r else "")
        raise NotImplementedError(f"HIP emitter: opcode {I.NAMES.get(typ, typ)} not supported")

    # statements ------------------------------------------------------------------------------------
    def group_body(self, g):
        """HIP statements of dependency group g (1-based)"""
        L = self.L
        out = [self.fused_sweeps(g), self.fused_dsweeps(g), self.exact_dsweeps(g)]
        self._rc = {}
        self._fw_rep, self._fw_fam = {}, {}
        self._pending = []          # slot ranges written by lane-split stores since the last wave barrier
        share = set()
        if self.recip:
            cnt = {}
            for i in L.T.groups[g - 1]:
                i = int(i)
                if L.inline[i] or L.in_family[i]:
                    continue
                for lf in self._den_leaves(L.tree(i), []):
                    cnt[lf.ins] = cnt.get(lf.ins, 0) + 1
            share = {j for j, c_ in cnt.items() if c_ >= 2}
        from .lowering import DFamily
        born = {}
        for uidx, u in enumerate(L.units[g - 1]):
            if self.fw_window:
                for key in [k_ for k_, b_ in born.items() if uidx - b_ > self.fw_window]:
                    self._fw_rep.pop(key, None)
                    self._fw_fam.pop(key, None)
                    del born[key]
            if isinstance(u, DFamily):
                out.append(self.dfamily_body(u))
                continue
            if isinstance(u, Family):
                e = self.expr(u.template, u)
                dst = f"{u.dst0}+c_" if u.ddst == 1 else f"{u.dst0}+({u.ddst})*c_"
                if L.P > 1 and self.forward and u.var == "c_":
                    nm = f"fw{u.dst0}"
                    nb = -(-u.n // L.P)
                    out.append("  double " + ", ".join(f"{nm}_{k}" for k in range(nb)) + ";\n")
                    self._barrier_if_needed(e, out)
                    for k in range(nb):
                        import re as _re
                        ek = _re.sub(r"@FW(\w+)@", lambda m_: f"{m_.group(1)}_{k}", e)
                        if (k + 1) * L.P <= u.n:
                            out.append(f"  {{ const int c_ = lane + {k * L.P}; {{ {nm}_{k} = {ek}; MM({dst}) = {nm}_{k}; }} }}\n")
                        elif not self.branch_free_tails or "SX(" in ek or "TC_SD(" in ek or "c.Sd" in ek:
                            # the members read global memory: lanes past the end must not (an index past the
                            # input block is an out-of-bounds HBM access, not a harmless LDS read)
                            out.append(f"  {{ const int c_ = lane + {k * L.P}; if (c_ < {u.n}) {{ {nm}_{k} = {ek}; "
                                       f"MM({dst}) = {nm}_{k}; }} }}\n")
                        else:
                            # ragged tail without a branch: the lanes past the end compute on whatever the
                            # next slots hold and park the result in a spare slot nobody reads
                            out.append(f"  {{ const int c_ = lane + {k * L.P}; {{ {nm}_{k} = {ek}; "
                                       f"MM((c_ < {u.n}) ? ({dst}) : TC_DUMMY) = {nm}_{k}; }} }}\n")
                    self._pending.append((u.dst0 + min(0, u.ddst * (u.n - 1)), u.dst0 + max(0, u.ddst * (u.n - 1))))
                    self._fw_fam[(u.dst0, u.ddst)] = (nm, u.n)
                    born[(u.dst0, u.ddst)] = uidx
                elif L.P > 1:
                    # the lanes sharing the instance split the members; LDS (in order within a
                    # wave) makes every element visible to all of them afterwards
                    out.append(self.lane_loop("c_", u.n, f"MM({dst}) = {e};") +
                               "  __builtin_amdgcn_wave_barrier();\n")
                else:
                    out.append(f"  for (int c_ = 0; c_ < {u.n}; c_++) {{ MM({dst}) = {e}; }}\n")
                continue
            i = int(u)
            if share:
                for lf in self._den_leaves(L.tree(i), []):
                    if lf.ins in share and lf.ins not in self._rc:
                        self._rc[lf.ins] = f"rc{lf.ins}"
                        if self.forward and L.P > 1:
                            self._barrier_if_needed(self.leaf_text(lf), out)
                        out.append(f"  const double rc{lf.ins} = TC_DIV(1.0,{self.leaf_text(lf)});\n")
            e = self.expr(L.tree(i))
            if self.forward and L.P > 1:
                self._barrier_if_needed(e, out)
            if L.local[i]:
                out.append(f"  const double t{i} = {e};\n")
            elif self.forward and L.P > 1:
                k_ = int(L.pslot[i])
                out.append(f"  const double r{k_}_ = {e};\n  MM({k_}) = r{k_}_;\n")
                if k_ in self._rv:
                    out.append(f"  c.rv[{self._rv[k_]}] = r{k_}_;\n")
                self._fw_rep[k_] = f"r{k_}_"
                born[k_] = uidx
            else:
                out.append(f"  MM({L.pslot[i]}) = {e};\n")
        self._fw_rep, self._fw_fam = {}, {}
        if self._pending:      # later groups / entry points read through LDS
            out.append("  __builtin_amdgcn_wave_barrier();\n")
            self._pending = []
        return "".join(out)

    # ---- register cache of hot replicated values (lanes>1) ----------------------------------------
    def setup_rcache(self, budget):
        """Replicated values (every lane of the instance computes the same number) that replicated code
        reads most often by constant index -- the pivots and factors of the sequential core of the LDL,
        the triangular solves, scalars -- are ALSO kept in a per-lane register (c.rv[j]) next to their LDS
        slot, so the sequential core runs without LDS round trips.  Only slots whose every writer is
        replicated code qualify (members of lane-split families and copy destinations do not).
        Returns the number of cached slots."""
        L, T = self.L, self.L.T
        if budget <= 0 or L.P == 1:
            return 0
        banned = set()
        for e in list(T.copies):
            for d in e["dst"]:
                if L.pslot[d] >= 0:
                    banned.add(int(L.pslot[d]))
        writer_ok = {}
        cnt = {}

        def visit(t, fam):
            for k in t.kids:
                if isinstance(k, Leaf):
                    if k.kind == "priv":
                        cnt[k.idx] = cnt.get(k.idx, 0) + 1      # refined below for family operands
                else:
                    visit(k, fam)
        for g in range(1, T.nGroups + 1):
            for u in L.units[g - 1]:
                if isinstance(u, Family):
                    # operands with stride 0 are broadcast reads of a constant slot
                    pos = 0
                    def walk(t):
                        nonlocal pos
                        for k in t.kids:
                            if isinstance(k, Leaf):
                                if k.kind == "priv" and u.delta[pos] == 0:
                                    cnt[u.base[pos]] = cnt.get(u.base[pos], 0) + 1
                                pos += 1
                            else:
                                walk(k)
                    walk(u.template)
                    for i in u.members:
                        banned.add(int(L.pslot[i]))
                elif isinstance(u, (int, np.integer)):
                    i = int(u)
                    visit(L.tree(i), None)
                    if not L.local[i] and L.pslot[i] >= 0:
                        writer_ok[int(L.pslot[i])] = True
        for e in T.sets:
            if not e["external"]:
                for i in e["ins"]:
                    if L.pslot[i] >= 0:
                        writer_ok[int(L.pslot[i])] = True
        cand = sorted((k for k in cnt if writer_ok.get(k) and k not in banned), key=lambda k: (-cnt[k], k))
        cand = [k for k in cand if cnt[k] >= 2][:int(budget)]
        self._rv = {k: j for j, k in enumerate(cand)}
        return len(cand)

    # ---- lazy wave barriers (forward mode) ----------------------------------------------------
    _MM_RE = None

    def _lds_reads(self, text):
        """slot ranges [(lo, hi)] a printed statement reads through LDS, None if it cannot be told"""
        import re
        if Printer._MM_RE is None:
            Printer._MM_RE = re.compile(r"MM\(([^()]*(?:\([^()]*\)[^()]*)*)\)")
        out = []
        for a in Printer._MM_RE.findall(text):
            a = a.replace(" ", "")
            m = re.fullmatch(r"(\d+)", a)
            if m:
                out.append((int(m.group(1)), int(m.group(1))))
                continue
            m = re.fullmatch(r"(\d+)\+(?:c_|j_)", a)
            if m:
                out.append((int(m.group(1)), int(m.group(1)) + 63))
                continue
            m = re.fullmatch(r"(\d+)\+\((-?\d+)\)\*(?:c_|j_)", a)
            if m:
                b_, d_ = int(m.group(1)), int(m.group(2))
                out.append((b_ + min(0, 63 * d_), b_ + max(0, 63 * d_)))
                continue
            return None
        return out

    def _barrier_if_needed(self, text, out):
        """a compiler-level wave barrier is needed only before a statement that reads, through LDS, a
        slot some OTHER lane of the instance may have written since the last barrier"""
        if not self._pending:
            return
        rd = self._lds_reads(text)
        if rd is None or any(lo <= phi and plo <= hi for lo, hi in rd for plo, phi in self._pending):
            out.append("  __builtin_amdgcn_wave_barrier();\n")
            self._pending = []

    def lane_loop(self, var, n, body):
        """members var = lane, lane+P, ... < n, fully unrolled: whole blocks of P members need no
        predicate, only a ragged tail does"""
        P_ = self.L.P
        out = ""
        for k in range(-(-n // P_)):
            guard = "" if (k + 1) * P_ <= n else f"if ({var} < {n}) "
            out += f"  {{ const int {var} = lane + {k * P_}; {guard}{{ {body} }} }}\n"
        return out

    def ref(self, i):
        return self.leaf_text(self.L.leaf(i))

    def group_uses_distributed(self, g):
        L = self.L
        for i in L.T.groups[g - 1]:
            i = int(i)
            if i in L.dval or i in L.dreduce:
                return True
        return False


def _mask(groups):
    m = 0
    for g in groups:
        m |= 1 << (g - 1)
    return m


def emit_class(spec, fp_contract="fast", block=64, stats=False, roll=True, min_waves=0, lanes=1,
               recip=False, fast_div=False, inline_max=0, reg_budget=150, scratch="auto", body="code",
               waves=1, fuse=False, static_clean=False, forward=False, rcache=0, lds_state=0, share=False,
               branch_free_tails=True, exact=False, stg_blocks=0, ilp=False, fw_window=0, sched_every=0,
               refresh_outputs=False):
    """Return the HIP source text of the solver class.

    exact=True (lanes>1, fp_contract='off', no recip/fast_div): the wave-cooperative lowering with the
    reference's arithmetic kept bit for bit (Printer.exact_dsweeps / lane_reduction): same outputs, status and
    iteration counts as the one-thread-per-instance `strict` lowering and the CPU oracle.

    branch_free_tails=False: ragged tails of lane-split families keep their predicate; saves the spare LDS slot
    and the tail pad the branch-free form needs (sls/w8s: exactly the difference between 6 and 7 wavefronts per CU).

    share=True: values of transient groups with disjoint life times inside a pass of the Newton loop share
    storage (Lowering._assign_shared_slots); such classes do not support tc_probe.

    lds_state=N (register-resident lowering only: lanes=1, roll=False): up to N long-lived state values
    per instance -- what depends on the parameters only and what is carried from one Newton iteration to
    the next -- live in LDS ([slot][thread], conflict-free) instead of competing with the transient values
    of an iteration for the 512 registers; N*8*block bytes of LDS per block set the occupancy.

    static_clean=True: tests and code of dependency groups that are certainly clean at an entry
    point's call sites in the Newton loop are left out there (loop_schedule.py); unelided twins
    ``<entry>_full`` serve the probe kernel and getOutputs.

    body="code": dependency groups are printed as HIP statements (the lowering of lowering.py).
    body="interp": the groups stay DATA -- a word stream walked in lock step by every lane of a
    wavefront (csrc/tc_interp.hip.inc, interp_schedule.py); lanes=1, state in HBM.  For tapes too
    large to print (tens of thousands of instructions).  waves=W: W wavefronts of one block share
    the 64 instances of a tile and split the records of every level between them.

    lanes=1: one thread per problem instance (generic lowering, bit-exact with fp_contract='off').
    lanes=P>1: P lanes cooperate on one instance -- large inputs and the element families built on
    them are distributed over the lanes (register-resident for the whole solve, read from HBM
    once), long sums become per-lane partial sums + a DPP butterfly all-reduce, everything else is
    computed redundantly by the P lanes on per-instance state held in LDS."""
    T = spec.tape
    interp = body == "interp"
    eq_solver = _solver_of(spec) == "ipmPDeq_CSsolver"
    # (printed code stays on the reference's groups: its slot reuse inside a group follows the reference's instruction order;
    # on the merged tape `strict` was measured slower -- 23.4 against 22.8 ms, mpc_dcmotor 32 768 -- and not bit-identical)
    if interp and not eq_solver and os.environ.get("TC_INTERP_FUSE", "1") != "0":
        # the tile machine walks the coarser groups of the LDS machine too (ldsm_schedule.fused_tape: fewer hand-overs,
        # joint level schedules; l1l2 16 384: 131.3 -> 127.9 ms); every value has a row of its own here, so any
        # partition into groups is safe
        from . import ldsm_schedule as lsch_
        T = lsch_.fused_tape(T, spec.defines)
    if eq_solver:
        # the static clean / dirty analysis and the life-time slot sharing restate the call structure of
        # ipmPD_CSsolver (loop_schedule.newton_program); the equilibrium loop keeps the reference's lazy flags
        static_clean = False
        share = False
        if lanes != 1:
            raise ValueError("equilibrium classes: one thread per instance (lanes=1) or body='interp' / 'ldsm'")
    if interp:
        if lanes != 1:
            raise ValueError("body='interp' needs lanes=1")
        scratch = "global"
        block = 64 * int(waves)
        L = Lowering(T, None, roll=False, lanes=1, spill_all=True)
    else:
        L = Lowering(T, getattr(spec, "code", None), roll=roll, lanes=lanes,
                     share=spec.defines if share else None)
    if block % lanes:
        raise ValueError("block must be a multiple of lanes")
    P = Printer(L, roll=roll, recip=recip, fuse=fuse)
    P.forward = bool(forward) and lanes > 1
    P.exact = bool(exact) and lanes > 1
    P.ilp = bool(ilp)
    P.fw_window = int(fw_window)
    P.sched_every = int(sched_every)
    if P.exact and (recip or fast_div or fuse or fp_contract != "off"):
        raise ValueError("exact=True needs fp_contract='off' and no recip/fast_div/fuse")
    P.branch_free_tails = bool(branch_free_tails)
    nrv = P.setup_rcache(rcache) if (P.forward and not interp) else 0
    if nrv:
        inline_max = 10 ** 9        # the register copies live in the kernel's context: every group inlined
    cls = spec.classname
    nW = max(1, (T.nGroups + 63) // 64)
    w = []
    w.append(f"// GENERATED by tenscalc_amd.hip_emitter for class '{cls}' -- do not edit.\n"
             f"// tape: {T.n} instructions, {T.nGroups} dependency groups, {L.nPriv} private doubles,"
             f" {L.nExt} input doubles per instance, {len(L.families)} rolled element families\n")
    w.append("#include <hip/hip_runtime.h>\n#include <math.h>\n#include <float.h>\n#include <stdint.h>\n\n")
    w.append(f'#define TC_CLASSNAME "{cls}"\n#define TC_BLOCK {int(block)}\n#define TC_P {int(lanes)}\n'
             f"#define TC_G (TC_BLOCK / TC_P)\n#define TC_NDIST {L.nDist}\n")
    lb = f"__launch_bounds__(TC_BLOCK, {int(min_waves)})" if min_waves else "__launch_bounds__(TC_BLOCK)"
    if os.environ.get("TC_WPE"):        # tuning aid (hashed into the build key): "1,1" -> amdgpu_waves_per_eu(1, 1)
        wpe_ = os.environ["TC_WPE"].split("-")
        lb += f" __attribute__((amdgpu_waves_per_eu({wpe_[0]}, {wpe_[-1]})))"
    w.append(f"#define TC_LAUNCH_BOUNDS {lb}\n")
    if stats:       # stats="phases": only the per-phase counters of the Newton loop (8 registers; the per-group
        # stats="trace": lane 0 of block 0 writes (tag, s_memtime) pairs to the dbg buffer at every group evaluation and
        # phase boundary (one pointer in registers: the least perturbing of the three)
        w.append("#define TC_STATS %d\n" % {"phases": 2, "trace": 3}.get(stats, 1))     # arrays perturb register allocation)
        w.append("#define TC_TRACE_BLOCK %d\n" % int(os.environ.get("TC_TRACE_BLOCK", "0")))
    w.append(_define_block(spec))
    nOut = sum(int(np.prod(sz)) if len(sz) else 1 for _, sz in spec.outputs)
    consts = []
    if interp:
        from . import interp_schedule as isch
        cs_interp = isch.class_stream(T, L, cls, int(waves))     # (the rows above the computed values: constants, split reductions)
        consts = cs_interp["consts"]
    dummy = 1 if (forward and lanes > 1 and branch_free_tails) else 0
    nmp = max(1, L.nPriv) + dummy                                  # + TC_DUMMY
    if P.exact and L.dreduce:                                      # staging area of the ordered sums (only tapes that sum
                                                                   # over lane-distributed operands have any: mpc_dcmotor at 32
                                                                   # lanes carried 1 056 unused slots = 8.4 KB per instance)
        # stg_blocks=0: as many row blocks per round as fit while 4 wavefronts per CU (one per SIMD: the
        # distributed inputs are register resident) still share the 160 KB of LDS
        P.stg_base = nmp
        per_blk = lanes * (lanes + 1)
        kmax = max([-(-v[1] // lanes) for v in L.dreduce.values()], default=1)
        room = (163840 // (4 * max(1, int(min_waves))) // (block // lanes) // 8 - 8 - nmp) // per_blk
        P.stg_blocks = int(stg_blocks) if stg_blocks else max(1, min(kmax, room))
        rmax = {}
        for i_, v in L.dreduce.items():
            rmax[int(T.group[i_])] = rmax.get(int(T.group[i_]), 0) + 1
        while P.stg_blocks * per_blk < max([-(-r // lanes) * lanes for r in rmax.values()], default=0):
            P.stg_blocks += 1                                      # the totals of a group's sums land there too
        nmp += P.stg_blocks * per_blk
    nmp += int(os.environ.get("TC_LDS_PAD_SLOTS", "0"))      # occupancy experiments only
    # per-instance stride = 8 mod 16 doubles: the 4 instances of a half-wave start 16 banks apart either way
    # (8 mod 32: 0,16,32,48; 24 mod 32: 0,48,32,16), so broadcast reads and 8-lane reads of consecutive
    # slots are conflict-free
    nmp += (8 - nmp) % 16
    w.append(f"#define TC_NM {max(1, L.nPriv + len(consts))}\n#define TC_NMP {nmp}\n#define TC_NEXT {L.nExt}\n#define TC_NOUT {nOut}\n"
             f"#define TC_NGROUPS {T.nGroups}\n#define TC_NSTW {nW}\n#define TC_DUMMY {max(1, L.nPriv)}\n#define TC_LDS_TAIL {8 if dummy else 0}\n\n")
    dmembers = ""
    # register budget for lane-distributed inputs (doubles per lane); larger inputs stay in HBM
    budget = reg_budget
    for d in sorted([x for x in L.inputs if x["dist"]], key=lambda x: x["rest"] * x["K"]):
        d["resident"] = d["rest"] * d["K"] <= budget
        if d["resident"]:
            budget -= d["rest"] * d["K"]
    for d in L.inputs:
        if d["dist"] and d["resident"]:
            dmembers += f"  double dX{d['arr']}[{d['rest'] * d['K']}];\n"
    if any(d["dist"] and not d["resident"] for d in L.inputs):
        w.append("#define TC_SD32 1\n")
    for f in L.dfamilies.values():
        dmembers += f"  double dF{f.fid}[{-(-f.n // L.P)}];\n"
    # where values that cross group boundaries live: per-thread private memory (lanes=1), LDS shared by
    # the lanes of an instance (lanes>1), or -- for tapes whose live state is too large for either --
    # a structure-of-arrays block in HBM, M[slot][B] (scratch="global", lanes=1 only)
    if scratch == "auto":
        scratch = "private" if lanes == 1 else "lds"
    if scratch == "global" and lanes != 1:
        raise ValueError("scratch='global' needs lanes=1")
    mem_t = {"private": "tc_priv", "lds": "tc_lds", "global": "tc_gmem"}[scratch]
    # LDS layout [instance-in-block][slot], slot count padded to 8 mod 32 doubles: the 8 lanes of an
    # instance reading 8 consecutive slots and the 4 instances of a half-wave then hit 32 distinct
    # 8-byte bank pairs (conflict-free ds_read_b64); a slot read by all lanes of an instance is a broadcast
    # HBM-resident state: one tile per wavefront, tile[row][64 instances] (rows of a tile contiguous)
    mm = "(m[k])" if scratch != "global" else "(m[(long)(k) * 64])"
    w.append("// per-instance state: m = scratch for values crossing group boundaries (private memory when\n"
             "// one thread owns the instance, LDS [slot][instance] when TC_P lanes share it), S = this\n"
             "// instance's column of the structure-of-arrays input block (element k at S[k*B]), dX*/dF* =\n"
             "// this lane's share of the distributed vectors (registers)\n"
             "typedef __attribute__((address_space(5))) double *tc_priv;\n"
             "typedef __attribute__((address_space(3))) double *tc_lds;\n"
             "typedef const __attribute__((address_space(1))) double *tc_glob;\n"
             "typedef __attribute__((address_space(1))) double *tc_gmem;\n"
             f"typedef {mem_t} tc_mem;\n#define TC_SCRATCH_{scratch.upper()} 1\n"
             "struct TcCtx {\n  tc_mem m;\n  tc_lds ls;\n  tc_glob S;\n  tc_glob Sd;\n  long B, sdB;\n  int lane;\n"
             "  const __attribute__((address_space(1))) char *Sdb;\n  unsigned sdo8, sdB8;\n"
             "  unsigned long long st[TC_NSTW];\n" + (f"  double rv[{nrv}];\n" if nrv else "") + dmembers +
             "#if defined(TC_STATS) && TC_STATS == 3\n  __attribute__((address_space(1))) double *trp, *trpEnd;   // trace cursor, end of the dbg buffer (stamps stop there)\n"
             "#elif defined(TC_STATS)\n  int nMinF;\n  long long cyc[TC_NGROUPS];\n  int cnt[TC_NGROUPS];\n  long long ph[8], tph;\n#endif\n"
             "#ifdef TC_VERBOSE3\n  double *v3, *v3cur;   // rows of the verbose table (ipm_newton.hip.inc: TC_V3); null on all but the traced instance\n"
             "  int v3n, v3max;\n#endif\n};\n"
             # input reads are clamped to the input block: every index the emitter prints is in range (constants
             # fold away), the clamp is the backstop for a lane of a split loop that should have been masked
             "#define SX(k) (S[(long)((k) < TC_NEXT ? (k) : (TC_NEXT > 0 ? TC_NEXT - 1 : 0)) * B])\n"
             f"#define MM(k) {mm}\n#define ML(j) (c.ls[(j) * TC_BLOCK])\n"
             "// element j of this lane's share of a distributed input that is not resident in registers;\n"
             "// 32-bit byte offsets: the host refuses batches whose distributed block reaches 4 GiB (TC_SD32)\n"
             "#define TC_SD(j) (*(tc_glob)(c.Sdb + (unsigned)((unsigned)((j) < TC_NDIST ? (j) : (TC_NDIST > 0 ? TC_NDIST - 1 : 0)) * c.sdB8 + c.sdo8)))\n"
             "#define TC_UNPACK tc_mem m = c.m; tc_glob S = c.S; const long B = c.B; const int lane = c.lane; (void)m; (void)S; (void)B; (void)lane;\n"
             "#define TC_NUM(x, d) (x)\n"
             + ("// opt-in fast division: hardware reciprocal + two Newton steps + one residual correction\n"
                "// (about 1 ulp, no denormal/overflow rescaling) instead of the correctly rounded sequence\n"
                "static __device__ __forceinline__ double tc_fdiv(double x, double d) {\n"
                "  double r = __builtin_amdgcn_rcp(d);\n  r = fma(fma(-d, r, 1.0), r, r);\n"
                "  r = fma(fma(-d, r, 1.0), r, r);\n  const double q = x * r;\n"
                "  return fma(fma(-d, q, x), r, q);\n}\n#define TC_DIV(x, d) tc_fdiv((x), (d))\n"
                if fast_div else "#define TC_DIV(x, d) ((x) / (d))\n") +
             "static __device__ __forceinline__ double tc_min(double a, double x) { return (a > x) ? x : a; }\n"
             "static __device__ __forceinline__ double tc_max(double a, double x) { return (a < x) ? x : a; }\n"
             "// I_clp step: if (dx<0) { t=-x/dx; if (a>t) a=t; }  (instructionsTableUTHash.c:1315-1331)\n"
             "static __device__ __forceinline__ double tc_clp(double a, double x, double dx) {\n"
             "  if (dx < 0) { const double t = -x / dx; if (a > t) a = t; }\n  return a;\n}\n"
             '#include "tc_wave.hip.inc"\n#include "tc_pow.hip.inc"\n\n')
    # dependency groups ------------------------------------------------------------------
    stream, copy_rng, out_rng = [], {}, None
    if interp:
        from . import interp_schedule as isch
        cs = cs_interp
        stream, granges, gstats = cs["dev_words"], cs["granges"], cs["gstats"]      # rows as byte offsets (device_words)
        copy_rng, out_rng, consts = cs["copy_rng"], cs["out_rng"], cs["consts"]
        arr = isch.as_array(stream + [0] * 8)          # the header prefetch looks one record past the end
        w.append(f"// the tape as data: {arr.size} words, {len(consts)} constants\n"
                 "static __device__ const int tc_code[] __attribute__((aligned(64))) = {\n")
        for k0 in range(0, arr.size, 32):
            w.append(",".join(str(int(x)) for x in arr[k0:k0 + 32]) + ",\n")
        # (the rows split reductions lead with hold +-inf: interp_schedule.split_long; the tape itself has no such constant)
        w.append("};\nstatic __device__ const double tc_consts[] = {"
                 + ", ".join(("-" if v < 0 else "") + "__builtin_inf()" if v in (float("inf"), float("-inf")) else _lit(v) for v in consts) + ("" if consts else "0.0") + "};\n"
                 "static __device__ const int tc_rng[] __attribute__((aligned(64))) = {"
                 + ", ".join(str(int(x)) for x in cs["table"]) + "};\n"
                 f"#define TC_NCONST {len(consts)}\n#define TC_NPRIV {L.nPriv}\n#define TC_W {int(waves)}\n"
                 "// the record shapes / opcodes of THIS stream (interp_schedule.shape_masks): nothing else is instantiated\n"
                 + "".join(f"#define TC_OPS_{k} 0x{v:x}ULL\n" for k, v in sorted(cs["shapes"].items()))
                 + '#define TC_OPS_GIVEN 1\n#include "tc_interp.hip.inc"\n\n')
        for g, members in enumerate(T.groups, 1):
            w.append(f"// group {g}: {len(members)} instructions in {gstats[g - 1][1]} levels, "
                     f"{int(T.flops[g].sum())} flops\n"
                     f"#define TC_CALL_GRP{g} tc_interp(c.m, {cs['rid'][g]})\n")
    for g, members in enumerate(T.groups, 1):
        if interp:
            break
        w.append(f"// group {g}: {len(members)} instructions, {int(T.flops[g].sum())} flops\n")
        if P.group_uses_distributed(g) or len(members) <= inline_max:
            # small groups are inlined at their call sites: a call costs more than they do
            w.append(f"static __device__ __forceinline__ void tc_grp{g}(TcCtx &c) {{\n  TC_UNPACK\n")
            w.append(P.group_body(g))
            w.append(f"}}\n#define TC_CALL_GRP{g} tc_grp{g}(c)\n")
        else:
            w.append(f"static __device__ __noinline__ void tc_grp{g}(tc_mem m, tc_glob S, const long B, const int lane) {{\n")
            w.append(P.group_body(g))
            w.append(f"}}\n#define TC_CALL_GRP{g} tc_grp{g}(c.m, c.S, c.B, c.lane)\n")
    w.append("\n// TC_EVAL: the group is certainly dirty here (loop_schedule.py): evaluate without testing the flag\n"
             "#define TC_EVAL(g) do { c.st[((g)-1)>>6] |= 1ull << (((g)-1)&63); TC_CALL_GRP##g; } while (0)\n")
    w.append("\n#if defined(TC_STATS) && TC_STATS == 3\n"
             "#define TC_STAMP(tag) do { if (c.trp && c.trp < c.trpEnd) { c.trp[0] = (double)(tag); c.trp[1] = (double)clock64(); c.trp += 2; } } while (0)\n"
             "#undef TC_EVAL\n"
             "#define TC_EVAL(g) do { c.st[((g)-1)>>6] |= 1ull << (((g)-1)&63); TC_STAMP(g); TC_CALL_GRP##g; TC_STAMP(-(g)); } while (0)\n"
             "#define TC_NEED(g) do { if (!((c.st[((g)-1)>>6] >> (((g)-1)&63)) & 1ull)) { "
             "c.st[((g)-1)>>6] |= 1ull << (((g)-1)&63); TC_STAMP(g); TC_CALL_GRP##g; TC_STAMP(-(g)); } } while (0)\n"
             "#elif !defined(TC_STATS) || TC_STATS == 2\n#define TC_NEED(g) do { if (!((c.st[((g)-1)>>6] >> (((g)-1)&63)) & 1ull)) { "
             "c.st[((g)-1)>>6] |= 1ull << (((g)-1)&63); TC_CALL_GRP##g; } } while (0)\n"
             "#else  // diagnostic build: wall cycles (s_memtime) and call count per dependency group\n"
             "#define TC_NEED(g) do { if (!((c.st[((g)-1)>>6] >> (((g)-1)&63)) & 1ull)) { "
             "c.st[((g)-1)>>6] |= 1ull << (((g)-1)&63); const long long t0_ = clock64(); TC_CALL_GRP##g; "
             "c.cyc[(g)-1] += clock64() - t0_; c.cnt[(g)-1]++; } } while (0)\n"
             "#undef TC_EVAL   // ... the executions the static schedule knows to be needed count too (the profile's totals)\n"
             "#define TC_EVAL(g) do { c.st[((g)-1)>>6] |= 1ull << (((g)-1)&63); const long long t0_ = clock64(); TC_CALL_GRP##g; "
             "c.cyc[(g)-1] += clock64() - t0_; c.cnt[(g)-1]++; } while (0)\n#endif\n")

    # Groups that use lane-distributed register state must be inlined wherever they are evaluated.
    # To keep ONE copy of their code per invalidation point instead of one per reader they are
    # PUSHED, not pulled: the set/copy that invalidates such a group re-evaluates it on the spot
    # (after its ancestors), so every get/copy can assume it is clean and never tests it.
    Dg = [g for g in range(1, T.nGroups + 1) if P.group_uses_distributed(g)]
    if static_clean and not interp:
        # ... except distributed groups that only getOutputs (or a get the loop never calls) needs: pushing
        # them would re-evaluate them -- for inputs not resident in registers: re-read the inputs from
        # HBM -- on every primal update although they are read once, after the loop; they are pulled there
        from .loop_schedule import newton_program

        def _names(nodes, acc):
            for nd in nodes:
                if nd[0] == "call":
                    acc.add(nd[1])
                elif nd[0] in ("if", "loop"):
                    _names(nd[1], acc)
            return acc
        called = _names(newton_program(spec.defines), set())
        in_loop = set()
        for e in list(T.gets) + list(T.copies):
            if e["name"] in called:
                in_loop |= set(e["parents"])
        Dg = [g for g in Dg if g in in_loop]
    gdeps = {g: set() for g in range(1, T.nGroups + 1)}
    for g, members in enumerate(T.groups, 1):
        for i in members:
            for j in L.operands(int(i)):
                gj = int(T.group[j])
                if gj and gj != g:
                    gdeps[g].add(gj)

    def ancestors(g, acc):
        for a in gdeps[g]:
            if a not in acc:
                acc.add(a)
                ancestors(a, acc)
        return acc

    def push(children):
        need = set()
        for g in children:
            if g in Dg:
                need.add(g)
                need |= ancestors(g, set())
        return "".join(f"  TC_NEED({g});\n" for g in sorted(need))

    internal_children = set()
    for e in list(T.sets) + list(T.copies):
        if not e.get("external"):
            internal_children |= set(e["children"])
    prologue_push = push([g for g in Dg if g not in internal_children])
    import re as _re
    pushed_first = sorted(int(x) for x in _re.findall(r"TC_NEED\((\d+)\)", prologue_push))

    def inval(children):
        s = ""
        for wd in range(nW):
            m = sum(1 << ((g - 1) % 64) for g in children if (g - 1) // 64 == wd)
            if m:
                s += f"  c.st[{wd}] &= ~0x{m:x}ull;\n"
        return s + push(children)

    # entry points ------------------------------------------------------------------------
    w.append("\n// ---- tape entry points (semantics: writeCfunctionpergroup.m:513-852)\n")
    for e in T.sets:
        if e["external"]:
            continue
        w.append(f"static __device__ __forceinline__ void {e['name']}(TcCtx &c, const double *input) {{\n  TC_UNPACK\n")
        if interp and int(waves) > 1:
            w.append("  tc_block_sync();\n")       # the other waves of the block may still be reading
        for k, i in enumerate(e["ins"]):
            if L.pslot[i] >= 0:
                w.append(f"  MM({L.pslot[i]}) = input[{k}];\n")
                if int(L.pslot[i]) in P._rv:
                    w.append(f"  c.rv[{P._rv[int(L.pslot[i])]}] = input[{k}];\n")
        w.append(inval(e["children"]))
        w.append("}\n")
    elide, sure = {}, {}
    if static_clean and not interp:
        from .loop_schedule import analyse
        elide, sure = analyse(T, spec.defines, pre_clean=pushed_first if (P.fuse and lanes > 1) else ())
    full = "_full" if elide else ""       # name suffix of the context-free twins

    def need(g, name, sfx):
        """evaluation of group g inside entry point `name`: nothing where it is certainly clean, no flag
        test where it is certainly dirty (TC_EVAL), the reference's lazy test otherwise"""
        if g in Dg or (not sfx and g in elide.get(name, ())):
            return ""
        if not sfx and g in sure.get(name, ()):
            return f"  TC_EVAL({g});\n"
        return f"  TC_NEED({g});\n"

    # Register families (lane-distributed element families: c.dF*) that getOutputs reads were last written by the push of
    # the final primal update; kept until then they are live across the whole Newton loop (sls: y = A*x - b, 26
    # registers of a kernel that spills).  Re-evaluating the pushed groups that DEFINE them right before getOutputs reads
    # them -- pure functions of the iterate, the same values -- makes them dead in between.
    fam_groups = {int(T.group[min(f_.members)]) for f_ in L.dfamilies.values()} if (refresh_outputs and lanes > 1) else set()
    for e in T.gets:
        args = ", ".join(f"double *y{k + 1}" for k in range(len(e["outputs"])))
        for sfx in ("", "_full") if elide else ("",):
            w.append(f"static __device__ __forceinline__ void {e['name']}{sfx}(TcCtx &c, {args}) {{\n  TC_UNPACK\n")
            if e["name"].endswith("_getOutputs") and fam_groups:
                anc = set()
                for g in e["parents"]:
                    anc |= ancestors(g, {g})
                for g in sorted(fam_groups & anc & set(Dg)):
                    w.append(f"  TC_CALL_GRP{g};\n")
            for g in e["parents"]:
                w.append(need(g, e["name"], "x" if not elide else sfx))
            for k, o in enumerate(e["outputs"]):
                for j, i in enumerate(o["ins"]):
                    w.append(f"  y{k + 1}[{j}] = {P.ref(i)};\n")
            w.append("}\n")
    for e, sfx in [(e, sfx) for e in T.copies for sfx in (("", "_full") if elide else ("",))]:
        w.append(f"static __device__ __forceinline__ void {e['name']}{sfx}(TcCtx &c) {{\n  TC_UNPACK\n")
        for g in e["parents"]:
            w.append(need(g, e["name"], "x" if not elide else sfx))
        pairs = []
        if interp:
            w.append(f"  tc_interp(c.m, {cs['rid'][e['name']]});\n")
            w.append(inval(e["children"]))
            w.append("}\n")
            continue
        for d, s_ in zip(e["dst"], e["src"]):
            if L.external[d]:
                raise NotImplementedError("copy into an external set")
            if L.pslot[d] >= 0:
                pairs.append((int(L.pslot[d]), L.leaf(s_)))
        k0 = 0
        while k0 < len(pairs):
            # maximal run with affine destination and affine LDS / input source: split over the lanes
            k1 = k0 + 1
            if lanes > 1 and pairs[k0][1].kind in ("priv", "ext") and k1 < len(pairs):
                dd = pairs[k1][0] - pairs[k0][0]
                ds = pairs[k1][1].idx - pairs[k0][1].idx if pairs[k1][1].kind == pairs[k0][1].kind else None
                while (k1 < len(pairs) and ds is not None and pairs[k1][1].kind == pairs[k0][1].kind
                       and pairs[k1][0] == pairs[k0][0] + dd * (k1 - k0)
                       and pairs[k1][1].idx == pairs[k0][1].idx + ds * (k1 - k0)):
                    k1 += 1
            if k1 - k0 >= 4:
                src = ("MM" if pairs[k0][1].kind == "priv" else "SX") + f"({pairs[k0][1].idx}+({ds})*j_)"
                w.append(P.lane_loop("j_", k1 - k0, f"MM({pairs[k0][0]}+({dd})*j_) = {src};") +
                         "  __builtin_amdgcn_wave_barrier();\n")
                k0 = k1
            else:
                w.append(f"  MM({pairs[k0][0]}) = {P.leaf_text(pairs[k0][1])};\n")
                k0 += 1
        w.append(inval(e["children"]))
        w.append("}\n")
    # entry points the loop may reference under #if but that this problem does not have
    if elide and lanes == 1 and scratch == "private" and not roll:
        # state promoted to registers: tell the compiler that what the groups certainly dirty at the head
        # of the Newton loop computed last iteration is dead (never read before it is recomputed)
        from .loop_schedule import analyse as _an
        slots = sorted({int(L.pslot[int(i)]) for g in _an.loop_head_dirty for i in T.groups[g - 1]
                        if L.pslot[int(i)] >= 0})
        w.append("#define TC_LOOP_HEAD_KILL { tc_mem m = c.m; " +
                 " ".join(f"m[{k}] = __builtin_nondeterministic_value(m[{k}]);" for k in slots) + " }\n")
    w.append('#include "%s"\n\n' % ("ipm_newton_eq.hip.inc" if eq_solver else "ipm_newton.hip.inc"))
    # kernels -------------------------------------------------------------------------------
    outs = T.entry(f"{cls}_getOutputs")[2]["outputs"]
    # common kernel prologue: thread -> (instance, lane), per-instance state, distributed inputs
    prologue = ("#ifdef TC_W  // TC_W waves of the block share the 64 instances of tile blockIdx.x\n"
                "  const long t_ = (long)blockIdx.x * 64 + (threadIdx.x & 63);\n#else\n"
                "  const long t_ = (long)blockIdx.x * TC_BLOCK + threadIdx.x;\n#endif\n"
                "  const long tid = t_ / TC_P;\n"
                "  if (tid >= B) return;\n"
                "  TcCtx c;\n"
                "#if defined(TC_SCRATCH_GLOBAL)\n  c.m = (tc_gmem)(Mg + ((tid >> 6) * (long)(TC_NEXT + TC_NM) + TC_NEXT) * 64 + (tid & 63));\n"
                "#elif TC_P == 1\n  double mm_[TC_NM];\n  c.m = (tc_priv)mm_;\n"
                "#ifdef TC_NLS\n  __shared__ double tc_lds_s[TC_NLS * TC_BLOCK];\n  c.ls = (tc_lds)&tc_lds_s[threadIdx.x];\n#endif\n"
                "#else\n  __shared__ double tc_lds_m[TC_NMP * TC_G + TC_LDS_TAIL];   // tail: reads past a ragged family tail of the last instance\n"
                "  c.m = (tc_lds)&tc_lds_m[(threadIdx.x / TC_P) * TC_NMP];\n#endif\n"
                "  c.lane = (int)(t_ % TC_P);\n  c.S = (tc_glob)(S + tid);\n  c.B = B;\n"
                "  for (int k = 0; k < TC_NSTW; k++) c.st[k] = 0;\n"
                "#if defined(TC_STATS) && TC_STATS == 3\n"
                "  c.trp = (dbg && blockIdx.x == TC_TRACE_BLOCK && threadIdx.x == 0) ? (__attribute__((address_space(1))) double *)dbg : nullptr;\n"
                "  c.trpEnd = c.trp ? c.trp + ((long)(2 * TC_NGROUPS + 10) * B - 4) : nullptr;   // the buffer tc_create allocated, minus the end record\n"
                "  if (c.trp) { c.trp[0] = 1000.0; c.trp[1] = (double)clock64(); c.trp += 2; }\n"
                "#elif defined(TC_STATS)\n  c.nMinF = 0;\n  for (int k = 0; k < TC_NGROUPS; k++) { c.cyc[k] = 0; c.cnt[k] = 0; }\n"
                "  for (int k = 0; k < 8; k++) c.ph[k] = 0;\n"
                "  const long long tstart_ = clock64();\n  c.tph = tstart_;\n#endif\n"
                "#ifdef TC_VERBOSE3   // dbg[0] = the instance whose iteration table is recorded, rows from dbg + 8 (tc_trace)\n"
                "  c.v3 = nullptr; c.v3cur = nullptr; c.v3n = 0; c.v3max = 0;\n"
                "  if (dbg && c.lane == 0 && tid == (long)dbg[0]) { c.v3 = dbg + 8; c.v3max = TC_V3_MAXROWS; }\n#endif\n")
    prologue += ("  c.Sd = (tc_glob)(Sd + tid * TC_P + c.lane);\n  c.sdB = B * TC_P;\n"
                 "  c.Sdb = (const __attribute__((address_space(1))) char *)Sd;\n"
                 "  c.sdo8 = (unsigned)((tid * TC_P + c.lane) * 8);\n  c.sdB8 = (unsigned)(B * TC_P * 8);\n")
    if interp:
        prologue += ("  for (int k = 0; k < TC_NCONST; k++) c.m[(long)(TC_NPRIV + k) * 64] = tc_consts[k];\n"
                     "  for (int k = 0; k < TC_NEXT; k++) c.m[(long)(k - TC_NEXT) * 64] = S[(long)k * B + tid];\n")
    for d in L.inputs:
        if d["dist"] and d["resident"]:
            cnt = d["rest"] * d["K"]
            prologue += (f"  _Pragma(\"unroll\") for (int j = 0; j < {cnt}; j++) "
                         f"c.dX{d['arr']}[j] = Sd[((long)({d['off']} + j) * B + tid) * TC_P + c.lane];\n")
    if prologue_push and P.fuse and lanes > 1 and pushed_first:
        # the parameter-only distributed groups, evaluated once per solve right here: their sums over inputs
        # that are not resident in registers share ONE sweep (A'A and A'b: one pass over A instead of two),
        # their bodies follow inline
        sweep = P.fused_dsweeps(pushed_first)
        P._prefused = dict(P._fused_d)
        prologue += "  { TC_UNPACK\n" + sweep
        for g_ in pushed_first:
            prologue += (f"  c.st[{(g_ - 1) >> 6}] |= 1ull << {(g_ - 1) & 63};\n" + P.group_body(g_))
        prologue += "  }\n"
        P._prefused = {}
    elif prologue_push:
        prologue += "  { TC_UNPACK\n" + prologue_push + "  }\n"
    outs = T.entry(f"{cls}_getOutputs")[2]["outputs"]
    w.append("extern \"C\" __global__ void TC_LAUNCH_BOUNDS\n"
             "tc_solve_kernel(const double *__restrict__ S, const double *__restrict__ Sd, double *__restrict__ O,\n"
             "                int *__restrict__ status, int *__restrict__ iter, double mu0, int maxIter, double ae0,\n"
             "                double ae1, long B, double *__restrict__ dbg, double *__restrict__ Mg) {\n" + prologue +
             "  int st = 0, it = 0;\n"
             "  TC_SOLVER_SYMBOL(c, mu0, maxIter, ae0, ae1, &st, &it);\n")
    off = 0
    out_meta = []
    if interp:
        w.append("  {\n")
        for g in T.entry(f"{cls}_getOutputs")[2]["parents"]:
            w.append(f"    TC_NEED({g});\n")
        w.append("    if (threadIdx.x >= 64) return;   // the other waves of the block hold the same results\n"
                 f"    tc_gather(c.m, c.B, {out_rng[0]}, {out_rng[1]}, O, tid);\n  }}\n")
        for k, o in enumerate(outs):
            out_meta.append((o["name"], o["size"], len(o["ins"]), off))
            off += len(o["ins"])
    else:
        # profiling builds: what the Newton loop alone executed (the reference's solve() does not evaluate the outputs;
        # the batched kernel does, in the same launch) -- rows [2G+10, 3G+10) of the stats buffer
        w.append("#if defined(TC_STATS) && TC_STATS == 1\n#define TC_STATS_LOOPCNT 1\n  int cntLoop_[TC_NGROUPS];\n"
                 "  for (int k = 0; k < TC_NGROUPS; k++) cntLoop_[k] = c.cnt[k];\n#endif\n")
        for k, o in enumerate(outs):
            w.append(f"  double y{k + 1}[{max(1, len(o['ins']))}];\n")
        w.append(f"  {cls}_getOutputs{full}(c, " + ", ".join(f"y{k + 1}" for k in range(len(outs))) + ");\n")
        w.append("  if (c.lane != 0) return;\n")
        for k, o in enumerate(outs):
            n_ = len(o["ins"])
            w.append(f"  for (int j = 0; j < {n_}; j++) O[(long)({off} + j) * B + tid] = y{k + 1}[j];\n")
            out_meta.append((o["name"], o["size"], n_, off))
            off += n_
    w.append("  status[tid] = st;\n  iter[tid] = it;\n"
             "#ifdef TC_VERBOSE3\n  if (c.v3) { dbg[1] = (double)c.v3n; dbg[2] = (double)st; dbg[3] = (double)it; }\n#endif\n"
             "#if defined(TC_STATS) && TC_STATS == 3\n  if (c.trp) { c.trp[0] = 1001.0; c.trp[1] = (double)clock64(); c.trp[2] = 0.0; }\n"
             "#elif defined(TC_STATS)\n  if (dbg) {\n    for (int k = 0; k < TC_NGROUPS; k++) {\n"
             "      dbg[(long)k * B + tid] = (double)c.cyc[k];\n      dbg[(long)(TC_NGROUPS + k) * B + tid] = (double)c.cnt[k];\n    }\n"
             "    dbg[(long)(2 * TC_NGROUPS) * B + tid] = (double)(clock64() - tstart_);\n"
             "    dbg[(long)(2 * TC_NGROUPS + 1) * B + tid] = (double)c.nMinF;\n"
             "    for (int k = 0; k < 8; k++) dbg[(long)(2 * TC_NGROUPS + 2 + k) * B + tid] = (double)c.ph[k];\n"
             "#ifdef TC_STATS_LOOPCNT\n    for (int k = 0; k < TC_NGROUPS; k++) dbg[(long)(2 * TC_NGROUPS + 10 + k) * B + tid] = (double)cntLoop_[k];\n#endif\n"
             "  }\n#endif\n}\n\n")
    # probe kernel: init sequence then one get
    w.append("extern \"C\" __global__ void __launch_bounds__(TC_BLOCK)\n"
             "tc_probe_kernel(const double *__restrict__ S, const double *__restrict__ Sd, double *__restrict__ out,\n"
             "                int which, double mu0, double ae0, double ae1, long B, double *__restrict__ Mg) {\n"
             "#if (defined(TC_STATS) && TC_STATS == 3) || defined(TC_VERBOSE3)\n  double *dbg = nullptr;\n#endif\n" + prologue
             + (("#if setAddEye2Hessian != 0\n  setAddEye2Hessian1__(c, &ae0);\n  setAddEye2Hessian2__(c, &ae1);\n#endif\n"
                 "  initPrimal__(c);\n#if scaleCost != 0\n  scaleCost__(c);\n#endif\n"
                 "#if nF > 0\n#if scaleInequalities != 0\n  scaleIneq__(c);\n#endif\n  setMu__(c, &mu0);\n  initDualIneq__(c);\n#endif\n"
                 "#if nG > 0\n  initDualEq__(c);\n#endif\n") if eq_solver else       # ipmPDeq_CSsolver.c:134-221
                ("#if setAddEye2Hessian != 0\n  setAddEye2HessianU__(c, &ae0);\n  setAddEye2HessianEq__(c, &ae1);\n#endif\n"
                 f"  initPrimal__{full}(c);\n"
                 f"#if scaleCost != 0\n  scaleCost__{full}(c);\n#endif\n"
                 f"#if nF > 0\n#if scaleInequalities != 0\n  scaleIneq__{full}(c);\n#endif\n  setMu__(c, &mu0);\n  initDualIneq__{full}(c);\n#endif\n"
                 f"#if nG > 0\n  initDualEqX__{full}(c);\n#endif\n")) +
             "  switch (which) {\n")
    probes = []
    for k, e in enumerate(T.gets if not share else []):     # shared storage has no schedule for a lone get
        tot = sum(len(o["ins"]) for o in e["outputs"])
        probes.append((e["name"], tot))
        w.append(f"    case {k}: {{\n")
        for q, o in enumerate(e["outputs"]):
            w.append(f"      double y{q + 1}[{max(1, len(o['ins']))}];\n")
        w.append(f"      {e['name']}{full}(c, " + ", ".join(f"y{q + 1}" for q in range(len(e["outputs"]))) + ");\n")
        po = 0
        for q, o in enumerate(e["outputs"]):
            n_ = len(o["ins"])
            w.append(f"      if (c.lane == 0) for (int j = 0; j < {n_}; j++) out[tid * {tot} + {po} + j] = y{q + 1}[j];\n")
            po += n_
        w.append("    } break;\n")
    w.append("    default: break;\n  }\n}\n\n")
    if lds_state and lanes == 1 and not roll and not interp:
        w = [_place_state_in_lds("".join(w), T, L, spec, int(lds_state))]
    w.append(_class_tail(spec, T, L, out_meta, probes, lanes, nOut, stats=stats))
    return "".join(w), L


def _ldsc_splittable(r, two_phase=True, split_hbm=True):
    """tc_ldsm.hip.inc, "two-phase records": SHORT records (those with operands / results in HBM too: the loads of a step
    are then in flight together); LDS-only ordered sums of up to TCL_FOLD_ALL (48) staged terms whose late terms, if any,
    follow the chain.  r = one row of the header table."""
    from . import instructions as I
    from . import ldsm_schedule as lsch
    kind, op, fl = int(r[0]) >> 8, int(r[0]) & 255, int(r[5])
    if not two_phase:
        return False
    if kind == 1:
        return split_hbm or not (fl & 3)
    if fl & 3:
        return False
    dots = (I.I_minus_dot, I.I_minus_dot_div, I.I_plus_minus_dot, I.I_plus_minus_dot_div)
    if kind == 3 and int(r[1]) <= 48 and (op in dots or op == lsch.F_SUM):
        nlm = (fl >> lsch.F_NLATE_SHIFT) & 3
        return nlm == 0 or (op in dots and bool(int(r[2]) & lsch.LATE_LAST))
    return False


def _ldsc_byte_words(cs, two_phase=True, split_hbm=True):
    """the code array of a compiled class: in the body of every LDS-only record that runs in two phases, operand words
    a*2+sign become a*8+sign and result / row addresses a become a*8 -- LDS BYTE addresses (the class has no static LDS:
    its dynamic LDS starts at 0), one `v_and` per operand instead of shift, mask and add of the base.  The stream itself
    (cs["words"]: what the CPU model and validate() read) is untouched."""
    import numpy as np
    from . import ldsm_schedule as lsch
    words = np.asarray(cs["words"], dtype=np.int64).copy()
    hd = np.asarray(cs["headers"], dtype=np.int64).reshape(-1, 8)
    for r in hd:
        kind, n, off, fl, cn = int(r[0]) >> 8, int(r[1]), int(r[4]), int(r[5]), int(r[6])
        if kind not in (1, 3) or (fl & 3) or not _ldsc_splittable(r, two_phase, split_hbm):
            continue
        nlm = (fl >> lsch.F_NLATE_SHIFT) & 3 if kind == 3 else 0
        na = 1 + n if kind == 1 else (12 if nlm else 5)
        S = -(-na // 4) * 4
        b = words[off:off + S * cn].reshape(cn, S)
        opw = lambda c_: ((b[:, c_] >> 1) << 3) | (b[:, c_] & 1)        # noqa: E731
        b[:, 0] = b[:, 0] << 3
        if kind == 1:
            for j in range(1, 1 + n):
                b[:, j] = opw(j)
        else:                 # dest, len, head0, head1, base (, nlate | positions, a0, b0, c0, a1, b1, c1)
            b[:, 2], b[:, 3] = opw(2), opw(3)
            b[:, 4] = b[:, 4] << 3
            if nlm:
                for j in range(6, 12):
                    b[:, j] = opw(j)
    return [int(x) for x in words]


def _ldsc_functions(cs, W, chunk=96, depth=12, trace=False):
    from . import instructions as I
    from . import ldsm_schedule as lsch
    two_phase = os.environ.get("TC_LDSC_TWO_PHASE", "1") != "0"
    wpe = "__attribute__((amdgpu_waves_per_eu(1, 1))) " if os.environ.get("TC_LDSC_WPE", "0") != "0" else ""
    split_hbm = os.environ.get("TC_LDSC_SPLIT_HBM", "1") != "0"     # tuning aids (hashed into the build key)
    sched_b = os.environ.get("TC_LDSC_SCHEDB", "0") != "0"          # measured: 202.3 ms with, 200.9 without (quadcopter, 512)
    """body="ldsm", compiled=True: the record stream printed as code.  Per record range (a dependency group or a copy
    entry point) and per wavefront of the workgroup: its records in stream order, each one a TCC_EXEC with the header
    fields as literals (csrc/tc_ldsm.hip.inc: tcl_exec_t folds to the record's own code), its body words requested
    `depth` records ahead into one of `depth` register sets, barriers where the stream has them; cut into functions of
    `chunk` records (clang's time is superlinear in the size of a function).  tc_ldsm_run becomes a switch.
    mpc_quadcopter, 512 instances, ms per batch: interpreted walk (ldsm8) 531; compiled, 1 / 2 / 4 / 8 wavefronts per
    instance at depth 4: 566 / 369 / 294 / 278; 4 wavefronts at depth 2 / 4 / 8 / 12 / 16: 323 / 294 / 276 / 275 / 277.
    On the final round-4 stream (steps a third as long): depth 6 / 8 / 12 / 16 / 20: 165.7 / 156.5 / 152.0 / 153.5 / 152.0 -> 12."""
    # functions stay at `chunk` = 96 records: a build with 384 per function (mpc_quadcopter, W = 4) took a GPU memory
    # access fault on its first launch in round 4.  Round 5 (scripts/ldsc_chunk_probe.py, offline): kernel descriptor,
    # scratch (196 vs 204 B) and registers of the two builds are the same; what differs is that functions beyond 32 767
    # dwords make LLVM RELAX branches into s_getpc / s_add / s_setpc sequences on a scavenged SGPR pair (36 of them in
    # the 35 k-instruction function, none in any 96-record build).  build_class therefore checks the CONDITION -- no
    # function of the walk within reach of that limit (_largest_function) -- instead of trusting the record count.
    depth = int(os.environ.get("TC_LDSC_DEPTH", depth))          # tuning aid (hashed into the build key)
    chunk = int(os.environ.get("TC_LDSC_CHUNK", chunk))          # diagnostic (ditto): scripts/ldsc_chunk_probe.py
    hd = np.asarray(cs["headers"], dtype=np.int64).reshape(-1, 8)
    table = cs["table"]
    nrid = len(table) // (2 * W)
    w = []
    calls = {}
    for rid in range(nrid):
        for k in range(W):
            h0, h1 = int(table[(rid * W + k) * 2]), int(table[(rid * W + k) * 2 + 1])
            recs = list(range(h0, h1))
            names = []
            for c0 in range(0, len(recs), chunk):
                part = recs[c0:c0 + chunk]
                nm = f"tcc_r{rid}_w{k}_{c0 // chunk}"
                names.append(nm)
                body = [f"static __device__ __noinline__ {wpe}void {nm}(tc_lds lds, tc_gmem gm, const int lane) {{\n"
                        "  tc_gcode_p code = (tc_gcode_p)tc_code;\n"
                        "  int " + ", ".join(f"R{q}[TCL_NR]" for q in range(depth)) + ";\n"]
                real = [h for h in part if (int(hd[h, 0]) >> 8) != 2]        # records with a body (not barriers)

                def lit(h):
                    r = hd[h]
                    return f"{int(r[0])}, {int(r[1])}, {int(r[2])}, {int(r[3])}, {int(r[4])}, {int(r[5]) | (int(r[6]) << 8)}"
                pos = {h: q for q, h in enumerate(real)}
                for q in range(min(depth, len(real))):
                    body.append(f"  TCC_LOAD(R{q % depth}, {lit(real[q])})\n")
                # steps of this wavefront: a record flagged "new level" opens one; a barrier record is one of its own;
                # at most `depth` records each (a record and the one `depth` behind it share a register set)
                steps, cur = [], None
                for h in part:
                    if (int(hd[h, 0]) >> 8) == 2:
                        steps.append([h])
                        cur = None
                        continue
                    if cur is None or (int(hd[h, 5]) & 4) or len(cur) >= depth:
                        cur = []
                        steps.append(cur)
                    cur.append(h)

                splittable = lambda h: _ldsc_splittable(hd[h], two_phase, split_hbm)      # noqa: E731
                for st in steps:
                    h0_ = st[0]
                    if (int(hd[h0_, 0]) >> 8) == 2:                              # a barrier record of its own
                        if trace and k == 0:
                            body.append(f"  TCC_STAMP({h0_ * 8})\n")
                        body.append(f"  tcc_barrier({'true' if int(hd[h0_, 5]) & 2 else 'false'});\n")
                        continue
                    if W == 1 and (int(hd[h0_, 5]) & 4):
                        # a new level: its records read what the level before wrote, through LDS, from OTHER lanes.  With
                        # several wavefronts the step's barrier orders that; a single wavefront needs no barrier in
                        # hardware (its LDS operations execute in order) but the COMPILER must not move a read above a
                        # write it cannot see the dependence of -- measured: without this, mpc_dcmotor at depth 8 walks
                        # 45 % of its instances into maxIter
                        body.append("  __builtin_amdgcn_fence(__ATOMIC_RELEASE, \"wavefront\");\n"
                                    "  __builtin_amdgcn_wave_barrier();\n"
                                    "  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, \"wavefront\");\n")
                    tr = trace and k == 0 and bool(int(hd[h0_, 5]) & 4)
                    if tr:
                        body.append(f"  TCC_STAMP({h0_ * 8})\n")      # stats="levels": when wavefront 0 enters the level
                    sp = [h for h in st if splittable(h)]
                    # folds first: theirs is the step's longest chain
                    sp.sort(key=lambda h: 0 if (int(hd[h, 0]) >> 8) == 3 else 1)
                    body.append("  {\n")
                    for h in sp:
                        body.append(f"    double G{pos[h] % depth}[TCL_NG];\n    TCC_GATHER(R{pos[h] % depth}, G{pos[h] % depth}, {lit(h)})\n")
                    for h in st:
                        if h not in sp:
                            body.append(f"    TCC_RUN(R{pos[h] % depth}, {lit(h)})\n")
                    if tr:       # ... has its operands (forced wait) / has stored its results / is through the barrier
                        body.append(f"    asm volatile(\"s_waitcnt lgkmcnt(0)\" ::: \"memory\");\n    TCC_STAMP({h0_ * 8 + 1})\n")
                    if sp and sched_b:
                        # every load of the step before any of its arithmetic: left alone the scheduler sinks the late
                        # operands' loads to their use, behind the chain -- one more LDS latency on the step's critical path
                        body.append("    __builtin_amdgcn_sched_barrier(0);\n")
                    for h in sp:
                        body.append(f"    TCC_FINISH(R{pos[h] % depth}, G{pos[h] % depth}, {lit(h)})\n")
                    if tr:
                        body.append(f"    asm volatile(\"s_waitcnt lgkmcnt(0)\" ::: \"memory\");\n    TCC_STAMP({h0_ * 8 + 2})\n")
                    body.append("  }\n")
                    bar = [h for h in st if int(hd[h, 5]) & 64]
                    if bar:
                        body.append(f"  tcc_barrier({'true' if any(int(hd[h, 5]) & 128 for h in bar) else 'false'});\n")
                        if tr:
                            body.append(f"  TCC_STAMP({h0_ * 8 + 3})\n")
                    for h in st:
                        q = pos[h]
                        if q + depth < len(real):
                            body.append(f"  TCC_LOAD(R{q % depth}, {lit(real[q + depth])})\n")
                body.append("}\n")
                w.append("".join(body))
            calls[(rid, k)] = names
    w.append("static __device__ __noinline__ void tc_ldsm_run(tc_lds lds, tc_gmem gm, const int rid) {\n"
             "  const int lane = (int)(threadIdx.x & 63);\n"
             "  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));\n"
             "  tc_block_sync();  // nobody may still be reading what this evaluation overwrites\n"
             f"  switch (__builtin_amdgcn_readfirstlane(rid) * {W} + wave) {{\n")
    for (rid, k), names in calls.items():
        if names:
            w.append(f"    case {rid * W + k}: " + " ".join(f"{nm}(lds, gm, lane);" for nm in names) + " break;\n")
    w.append("    default: break;\n  }\n}\n\n")
    return "".join(w)


def emit_class_ldsm(spec, waves=4, stats=False, lds_slots=20000, fuse_chain=True, compiled=False, lds_stage=0, min_waves=0,
                    lds_split_min=0):
    """body="ldsm": ONE instance per workgroup of `waves` wavefronts, hot state in LDS, the tape as data whose
    records give every lane a different scalar instruction (csrc/tc_ldsm.hip.inc, ldsm_schedule.py).
    min_waves=k: the kernel is compiled for k wavefronts per SIMD (512 / k registers), so that as many workgroups as the LDS
    budget allows are really resident -- 4 instances x 2 wavefronts per CU need 2 per SIMD."""
    from . import ldsm_schedule as lsch
    if _solver_of(spec) == "ipmPDeq_CSsolver":
        raise NotImplementedError("body='ldsm' packs the state by life times derived from the call structure of "
                                  "ipmPD_CSsolver; equilibrium classes: lanes=1 printed code or body='interp'")
    T = spec.tape
    if fuse_chain:
        # the class walks a tape whose dependency groups are coarser than the reference's (ldsm_schedule.fused_tape: the
        # factorisation and the substitutions that read it as ONE level schedule; the groups one solve cannot tell apart
        # as one group); values are unchanged
        T = lsch.fused_tape(T, spec.defines, split_div=os.environ.get("TC_LDSM_SPLIT_DIV", "0") == "1")
    cls = spec.classname
    W = int(waves)
    # lds_split_min: long minima / maxima / sums of products with more terms than this are folded in two levels (default
    # ldsm_schedule.SPLIT_MIN = 96).  mpc_dcmotor's 96-term cost row sat exactly on the default: one lane folded it alone, and
    # the compiled walk requested the whole row at once -- 192 registers, spilled under the 248-register cap of `ldscs`
    # (3.2 GB of scratch writes per launch); with 48 it is two independent lines of 50 + 46 terms (the reference's own
    # statement split, instructionsTableUTHash.c:1003-1012): 19.9 -> 18.95 ms, bit-identical (profiles/r06_dcmotor_split_min.log)
    cs = lsch.class_stream(T, cls, spec.defines, W=W, lds_slots=lds_slots, stage=lds_stage, split=int(lds_split_min) or True)
    L = cs["lowering"]
    addr = cs["addr"]
    nW = max(1, (T.nGroups + 63) // 64)
    w = []
    w.append(f"// GENERATED by tenscalc_amd.hip_emitter (ldsm back-end) for class '{cls}' -- do not edit.\n"
             f"// tape: {T.n} instructions, {T.nGroups} dependency groups; state: {cs['nLds']} LDS slots "
             f"({cs['nHot']} hot shared, {cs['nLocal']} group-local scratch), {cs['nGlob']} doubles in HBM per instance "
             f"({cs['nExt']} inputs, {cs['nCold']} cold); {len(cs['headers']) // 8} records, {len(cs['words'])} body words\n")
    wpe_ = os.environ.get("TC_WPE") or (str(int(min_waves)) if min_waves else "")     # TC_WPE: tuning aid (hashed into the key)
    w.append("#include <hip/hip_runtime.h>\n#include <math.h>\n#include <float.h>\n#include <stdint.h>\n\n")
    w.append(f'#define TC_CLASSNAME "{cls}"\n#define TC_BLOCK {64 * W}\n#define TC_P 1\n#define TC_G TC_BLOCK\n'
             f"#define TC_NDIST 0\n#define TC_LDSM 1\n#define TC_W {W}\n#define TC_NLDS {cs['nLds']}\n#define TC_NGLOB {max(1, cs['nGlob'])}\n"
             "#define TC_LAUNCH_BOUNDS __launch_bounds__(TC_BLOCK)"
             + (" __attribute__((amdgpu_waves_per_eu({0}, {1})))".format(*(wpe_.split("-") * 2)[:2]) if wpe_ else "") + "\n")
    if stats:
        # stats="levels" (compiled walk only): wavefront 0 of block 0 writes (record, clock) into the dbg buffer whenever
        # it enters a level; the cursor lives in the last padding slot of the staging area (scripts/level_trace.py)
        w.append("#define TC_STATS %d\n" % ({"phases": 2, "levels": 4}.get(stats, 1)))
        if stats == "levels":
            w.append(f"#define TCC_CUR_SLOT {int(cs['stage_base']) + int(cs['nStage']) + 7}   // read padding of the staging area: never written\n")
    w.append(_define_block(spec))
    nOut = cs["nOut"]
    w.append(f"#define TC_NM {cs['nLds']}\n#define TC_NMP {cs['nLds']}\n#define TC_NEXT {cs['nExt']}\n#define TC_NOUT {nOut}\n"
             f"#define TC_NGROUPS {T.nGroups}\n#define TC_NSTW {nW}\n\n")
    w.append("typedef __attribute__((address_space(3))) double *tc_lds;\n"
             "typedef const __attribute__((address_space(1))) double *tc_glob;\n"
             "typedef __attribute__((address_space(1))) double *tc_gmem;\n"
             "// per-instance context: m = the instance's LDS state, gm = its HBM array (inputs first); every lane of the\n"
             "// block holds the same scalars and takes the same branches\n"
             "struct TcCtx {\n  tc_lds m;\n  tc_gmem gm;\n  long B;\n  int lane;\n  unsigned long long st[TC_NSTW];\n"
             "#ifdef TC_STATS\n  int nMinF;\n  long long cyc[TC_NGROUPS];\n  int cnt[TC_NGROUPS];\n  long long ph[8], tph;\n#endif\n};\n"
             "#define TC_DIV(x, d) ((x) / (d))\n"
             "static __device__ __forceinline__ double tc_min(double a, double x) { return (a > x) ? x : a; }\n"
             "static __device__ __forceinline__ double tc_max(double a, double x) { return (a < x) ? x : a; }\n"
             "static __device__ __forceinline__ double tc_clp(double a, double x, double dx) {\n"
             "  if (dx < 0) { const double t = -x / dx; if (a > t) a = t; }\n  return a;\n}\n\n")
    code_words = cs["words"]
    if compiled and os.environ.get("TC_LDSC_BYTES", "1") != "0":
        code_words = _ldsc_byte_words(cs, os.environ.get("TC_LDSC_TWO_PHASE", "1") != "0", os.environ.get("TC_LDSC_SPLIT_HBM", "1") != "0")
        w.append("#define TC_LDSC_BYTES 1\n")
    arr = lsch.as_array(code_words + [0] * 1024)    # the body prefetch reads up to 16 rows of 64 past a record
    w.append(f"// the tape as data: {arr.size} body words, {len(cs['headers']) // 8} record headers\n"
             "static __device__ const int tc_code[] __attribute__((aligned(64))) = {\n")
    for k0 in range(0, arr.size, 32):
        w.append(",".join(str(int(x)) for x in arr[k0:k0 + 32]) + ",\n")
    # header fields structure-of-arrays: [w0, n, aux, count, body offset, flags | cn << 8][record]
    hd = np.asarray(cs["headers"], dtype=np.int64).reshape(-1, 8)
    nrec = max(1, hd.shape[0])
    hf = np.zeros((6, nrec), dtype=np.int64)
    if hd.shape[0]:
        hf[0], hf[1], hf[2], hf[3], hf[4] = hd[:, 0], hd[:, 1], hd[:, 2], hd[:, 3], hd[:, 4]
        hf[5] = hd[:, 5] | (hd[:, 6] << 8)
    harr = lsch.as_array(list(hf.reshape(-1)))
    w.append(f"}};\n#define TC_NREC {nrec}\nstatic __device__ const int tc_hf[] __attribute__((aligned(64))) = {{\n")
    for k0 in range(0, harr.size, 32):
        w.append(",".join(str(int(x)) for x in harr[k0:k0 + 32]) + ",\n")
    w.append("};\nstatic __device__ const int tc_rng[] __attribute__((aligned(64))) = {"
             + ", ".join(str(int(x)) for x in cs["table"]) + ", 0, 0};\n"
             "static __device__ const double tc_consts[] = {" + ", ".join(
                 ("__builtin_inf()" if v > 0 else "-__builtin_inf()") if v in (float("inf"), float("-inf")) else _lit(v)
                 for _, v in cs["consts"]) + "};\n"
             "static __device__ const int tc_const_addr[] = {" + ", ".join(str(int(a)) for a, _ in cs["consts"]) + "};\n"
             f"#define TC_NCONST {len(cs['consts'])}\n"
             + ("#define TC_LDSC 1\n" if compiled else "") +
             '#include "tc_ldsm.hip.inc"\n\n')
    if compiled:
        if stats == "levels":
            w.append("// (cursor, buffer and its size live in three padding slots of the staging area: a stamp is two LDS reads, a\n"
                     "// clock read, one store -- no global load)\n"
                     "#define TCC_STAMP(h) if (blockIdx.x == 0 && lane == 0) { \\\n"
                     "    __attribute__((address_space(3))) long long *q_ = (__attribute__((address_space(3))) long long *)&lds[TCC_CUR_SLOT - 2]; \\\n"
                     "    const long long n_ = q_[2], lim_ = q_[1]; double *b_ = (double *)q_[0]; \\\n"
                     "    if (n_ < lim_) { b_[2 * n_] = (double)(h); b_[2 * n_ + 1] = (double)clock64(); q_[2] = n_ + 1; } }\n")
        w.append(_ldsc_functions(cs, W, trace=(stats == "levels")))
    for g, members in enumerate(T.groups, 1):
        w.append(f"// group {g}: {len(members)} instructions in {cs['gstats'][g][1]} levels, {int(T.flops[g].sum())} flops\n"
                 f"#define TC_CALL_GRP{g} tc_ldsm_run(c.m, c.gm, {cs['rid'][g]})\n")
    w.append("\n// lazy evaluation (writeCfunctionpergroup.m:44,653-657) of the groups in a parent mask, lowest group first: a loop\n"
             "// over the dirty bits instead of one flag test + call site per parent (record range of group g = g-1)\n"
             "static __device__ __forceinline__ void tc_need_mask(TcCtx &c, const int wd, unsigned long long parents) {\n"
             "  unsigned long long m = parents & ~c.st[wd];\n"
             "  while (m) {\n    const int b = __builtin_ctzll(m);\n    m &= m - 1;\n    c.st[wd] |= 1ull << b;\n"
             "#if defined(TC_STATS) && TC_STATS == 1\n    const long long t0_ = clock64();\n#endif\n"
             "    tc_ldsm_run(c.m, c.gm, wd * 64 + b);\n"
             "#if defined(TC_STATS) && TC_STATS == 1\n    c.cyc[wd * 64 + b] += clock64() - t0_;\n    c.cnt[wd * 64 + b]++;\n#endif\n"
             "  }\n}\n")

    def need(parents):
        s_ = ""
        for wd in range(nW):
            m = sum(1 << ((g - 1) % 64) for g in parents if (g - 1) // 64 == wd)
            if m:
                s_ += f"  tc_need_mask(c, {wd}, 0x{m:x}ull);\n"
        return s_

    def inval(children):
        s_ = ""
        for wd in range(nW):
            m = sum(1 << ((g - 1) % 64) for g in children if (g - 1) // 64 == wd)
            if m:
                s_ += f"  c.st[{wd}] &= ~0x{m:x}ull;\n"
        return s_

    def ref(i):
        i = int(i)
        if T.isload[i - 1]:
            return _lit(float(T.pars[T.parStart[i - 1]]))
        return f"TCL_LD({int(addr[i])})"
    w.append("\n// ---- tape entry points (semantics: writeCfunctionpergroup.m:513-852)\n")
    for e in T.sets:
        if e["external"]:
            continue
        w.append(f"static __device__ __forceinline__ void {e['name']}(TcCtx &c, const double *input) {{\n"
                 "  tc_block_sync();\n")        # the other waves of the block may still be reading
        for k, i in enumerate(e["ins"]):
            if addr[i] >= 0:
                w.append(f"  TCL_ST({int(addr[i])}, input[{k}]);\n")
        w.append(inval(e["children"]) + "}\n")
    for e in T.gets:
        args = ", ".join(f"double *y{k + 1}" for k in range(len(e["outputs"])))
        w.append(f"static __device__ __forceinline__ void {e['name']}(TcCtx &c, {args}) {{\n")
        w.append(need(e["parents"]))
        if e["name"].endswith("_getOutputs"):
            w.append("  tc_block_sync();\n}\n")      # the kernel gathers the outputs itself
            continue
        w.append("  tc_block_sync();\n")
        for k, o in enumerate(e["outputs"]):
            for j, i in enumerate(o["ins"]):
                w.append(f"  y{k + 1}[{j}] = {ref(i)};\n")
        w.append("}\n")
    for e in T.copies:
        w.append(f"static __device__ __forceinline__ void {e['name']}(TcCtx &c) {{\n")
        w.append(need(e["parents"]))
        w.append(f"  tc_ldsm_run(c.m, c.gm, {cs['rid'][e['name']]});\n" + inval(e["children"]) + "}\n")
    w.append('\n#include "%s"\n\n' % ("ipm_newton_eq.hip.inc" if _solver_of(spec) == "ipmPDeq_CSsolver" else "ipm_newton.hip.inc"))
    outs = T.entry(f"{cls}_getOutputs")[2]["outputs"]
    gargs = ", ".join("(double *)0" for _ in outs)
    w.append("extern \"C\" __global__ void TC_LAUNCH_BOUNDS\n"
             "tc_solve_kernel(const double *__restrict__ S, const double *__restrict__ Sd, double *__restrict__ O,\n"
             "                int *__restrict__ status, int *__restrict__ iter, double mu0, int maxIter, double ae0,\n"
             "                double ae1, long B, double *__restrict__ dbg, double *__restrict__ Mg) {\n"
             "  extern __shared__ double tc_lds_dyn[];\n"
             "  const long tid = (long)blockIdx.x;\n  if (tid >= B) return;\n"
             "  TcCtx c;\n  c.m = (tc_lds)tc_lds_dyn;\n  c.gm = (tc_gmem)(Mg + tid * (long)TC_NGLOB);\n  c.B = B;\n"
             "  c.lane = (int)(threadIdx.x & 63);\n  for (int k = 0; k < TC_NSTW; k++) c.st[k] = 0;\n"
             "#ifdef TC_STATS\n  c.nMinF = 0;\n  for (int k = 0; k < TC_NGROUPS; k++) { c.cyc[k] = 0; c.cnt[k] = 0; }\n"
             "  for (int k = 0; k < 8; k++) c.ph[k] = 0;\n  const long long tstart_ = clock64();\n  c.tph = tstart_;\n#endif\n"
             "#ifdef TC_LDSC_BYTES\n  // two-phase records address the LDS by byte offsets from 0: the class has no static LDS\n"
             "  if ((unsigned)(size_t)c.m != 0u) { if (threadIdx.x == 0) { status[tid] = 0x7fffffff; iter[tid] = -1; } return; }\n#endif\n"
             "  for (int k = threadIdx.x; k < TC_NCONST; k += TC_BLOCK) c.m[tc_const_addr[k]] = tc_consts[k];\n"
             "  if (threadIdx.x == 0) c.m[0] = 0.0;\n"
             + ("  if (threadIdx.x == 0) { __attribute__((address_space(3))) long long *q_ = (__attribute__((address_space(3))) long long *)&c.m[TCC_CUR_SLOT - 2]; "
                "q_[0] = (long long)dbg; q_[1] = ((long long)(2 * TC_NGROUPS + 10) * B) / 2; q_[2] = 0; }\n"
                if stats == "levels" else "") +
             "  __syncthreads();\n"
             "  int st = 0, it = 0;\n"
             "  TC_SOLVER_SYMBOL(c, mu0, maxIter, ae0, ae1, &st, &it);\n"
             f"  {cls}_getOutputs(c, {gargs});\n"
             f"  for (int j = threadIdx.x; j < TC_NOUT; j += TC_BLOCK) {{ const int a = tc_code[{cs['out_off']} + j]; "
             "O[(long)j * B + tid] = TCL_LD(a); }\n"
             "  if (threadIdx.x == 0) { status[tid] = st; iter[tid] = it; }\n"
             + ("#if 0   // stats=\"levels\": the dbg buffer holds the stamps\n" if stats == "levels" else "#ifdef TC_STATS\n") +
             "  if (dbg && threadIdx.x == 0) {\n"
             "    for (int k = 0; k < TC_NGROUPS; k++) {\n"
             "      dbg[(long)k * B + tid] = (double)c.cyc[k];\n      dbg[(long)(TC_NGROUPS + k) * B + tid] = (double)c.cnt[k];\n    }\n"
             "    dbg[(long)(2 * TC_NGROUPS) * B + tid] = (double)(clock64() - tstart_);\n"
             "    dbg[(long)(2 * TC_NGROUPS + 1) * B + tid] = (double)c.nMinF;\n"
             "    for (int k = 0; k < 8; k++) dbg[(long)(2 * TC_NGROUPS + 2 + k) * B + tid] = (double)c.ph[k];\n  }\n#endif\n}\n\n"
             "// no probe entry points in this back-end (storage is shared by life time)\n"
             "extern \"C\" __global__ void __launch_bounds__(TC_BLOCK)\n"
             "tc_probe_kernel(const double *__restrict__ S, const double *__restrict__ Sd, double *__restrict__ out,\n"
             "                int which, double mu0, double ae0, double ae1, long B, double *__restrict__ Mg) {}\n\n")
    off = 0
    out_meta = []
    for o in outs:
        out_meta.append((o["name"], o["size"], len(o["ins"]), off))
        off += len(o["ins"])
    w.append(_class_tail(spec, T, L, out_meta, [], 1, nOut, stats=stats))
    return "".join(w), L


def _place_state_in_lds(src, T, L, spec, budget):
    """register-resident lowering: move the long-lived slots to LDS (textual: every state access of this
    lowering is MM(<constant>))"""
    import re
    from .loop_schedule import analyse
    analyse(T, spec.defines)
    transient = set(analyse.loop_head_dirty)
    # who reads each value inside the loop: instructions of transient groups, loop entry points directly
    read_in_loop = set()
    for i in range(1, T.n + 1):
        if T.used[i] and int(T.group[i]) in transient:
            for j in L.operands(i):
                read_in_loop.add(int(j))
    for e in list(T.gets) + list(T.copies):
        if e["name"].endswith("_getOutputs") or e["name"] in ("getGrad__", "getJ__", "getHessInertia__"):
            continue
        for j in list(e.get("src", [])) + [x for o in e.get("outputs", []) for x in o["ins"]]:
            read_in_loop.add(int(j))
    longlived = set()
    for i in range(1, T.n + 1):
        k = int(L.pslot[i])
        if k < 0:
            continue
        g_ = int(T.group[i])
        if (g_ == 0 or g_ not in transient) and i in read_in_loop:
            longlived.add(k)             # carried between iterations, or parameter-only and used every iteration
    if re.search(r"MM\((?!\d+\))", src.split("#define MM(k)")[1].split("\n", 1)[1]):
        raise AssertionError("lds_state needs constant state indices everywhere (roll=False)")
    refs = {}
    for m_ in re.finditer(r"MM\((\d+)\)", src):
        refs[int(m_.group(1))] = refs.get(int(m_.group(1)), 0) + 1
    cand = sorted((k for k in longlived if refs.get(k, 0) > 0), key=lambda k: (refs[k], k))[:budget]
    where = {k: j for j, k in enumerate(cand)}
    head, body = src.split("#define MM(k)", 1)
    body = re.sub(r"MM\((\d+)\)", lambda m_: f"ML({where[int(m_.group(1))]})" if int(m_.group(1)) in where else m_.group(0), body)
    return head + f"#define TC_NLS {max(1, len(cand))}\n#define MM(k)" + body


def _class_tail(spec, T, L, out_meta, probes, lanes, nOut, stats=False):
    """metadata tables + reference-name wrappers + the host runtime (shared by every back-end)"""
    cls = spec.classname
    w = []
    # metadata tables ---------------------------------------------------------------------------
    isvar = {nm for nm, _ in spec.variables}

    def carr(ctype, name, vals, fmt=str):
        return f"static const {ctype} {name}[] = {{{', '.join(fmt(v) for v in vals) or '0'}}};\n"

    def dims8(sz):
        d = list(sz) + [1] * (8 - len(sz))
        return "{" + ", ".join(str(int(x)) for x in d) + "}"

    ins = L.inputs
    w.append(f"static const int TC_NIN = {len(ins)};\n")
    w.append(carr("char *const", "TC_IN_NAME", [i["name"] for i in ins], lambda s: f'"{s}"'))
    w.append(carr("int", "TC_IN_ISVAR", [int(i["name"] in isvar) for i in ins]))
    w.append(carr("int64_t", "TC_IN_NUMEL", [i["numel"] for i in ins]))
    w.append(carr("int64_t", "TC_IN_OFF", [i["off"] for i in ins]))
    w.append(carr("int", "TC_IN_DIST", [int(i["dist"]) for i in ins]))
    w.append(carr("int64_t", "TC_IN_N1", [i.get("n1", 0) for i in ins]))
    w.append(carr("int64_t", "TC_IN_K", [i.get("K", 0) for i in ins]))
    w.append(carr("int64_t", "TC_IN_REST", [i.get("rest", 0) for i in ins]))
    w.append(carr("int", "TC_IN_NDIM", [len(i["size"]) for i in ins]))
    w.append("static const int64_t TC_IN_DIMS[][8] = {" + ", ".join(dims8(i["size"]) for i in ins) + "};\n")
    w.append(f"static const int TC_NOUTS = {len(out_meta)};\n")
    w.append(carr("char *const", "TC_OUT_NAME", [o[0] for o in out_meta], lambda s: f'"{s}"'))
    w.append(carr("int64_t", "TC_OUT_NUMEL", [o[2] for o in out_meta]))
    w.append(carr("int64_t", "TC_OUT_OFF", [o[3] for o in out_meta]))
    w.append(carr("int", "TC_OUT_NDIM", [len(o[1]) for o in out_meta]))
    w.append("static const int64_t TC_OUT_DIMS[][8] = {" + ", ".join(dims8(o[1]) for o in out_meta) + "};\n")
    dn = sorted(spec.defines)
    w.append(f"static const int TC_NDEF = {len(dn)};\n")
    w.append(carr("char *const", "TC_DEF_NAME", dn, lambda s: f'"{s}"'))
    w.append(carr("double", "TC_DEF_VAL", [spec.defines[k] for k in dn], lambda v: repr(float(v))))
    w.append(f"static const int TC_NPROBE = {len(probes)};\n")
    w.append(carr("char *const", "TC_PROBE_NAME", [p[0] for p in probes], lambda s: f'"{s}"'))
    w.append(carr("int64_t", "TC_PROBE_NUMEL", [p[1] for p in probes]))
    stat = {"total_instructions": int(T.n), "used_instructions": int(T.used.sum()),
            "groups": int(T.nGroups), "private_doubles": int(L.nPriv), "input_doubles": int(sum(i["numel"] for i in ins)),
            "lanes_per_instance": int(lanes), "distributed_families": len(L.dfamilies),
            "distributed_reductions": len(L.dreduce), "rolled_families": len(L.families),
            "output_doubles": int(nOut), "flops_all_groups": int(T.flops.sum())}
    for g in range(1, T.nGroups + 1):
        stat[f"flops_group{g}"] = int(T.flops[g].sum())
    if stats and {"phases": 2, "trace": 3, "levels": 4}.get(stats, 1) == 1:
        # operations by class per dependency group (Tape._count_flops = the accounting of writeCinstructionsC): the
        # profiling view turns the execution counts of a solve into the reference's "Operation Count" table
        w.append("static const int64_t TC_GRP_FLOPS[][17] = {"
                 + ", ".join("{" + ", ".join(str(int(x)) for x in T.flops[g]) + "}" for g in range(1, T.nGroups + 1)) + "};\n")
        stat["reference_scratchbook"] = int(T.reference_memory_map()[1])
    w.append(f"static const int TC_NSTAT = {len(stat)};\n")
    w.append(carr("char *const", "TC_STAT_NAME", list(stat), lambda s: f'"{s}"'))
    w.append(carr("int64_t", "TC_STAT_VAL", list(stat.values())))
    # reference-exact single-instance wrappers (B=1)
    cw = ""
    for i in ins:
        cw += (f'void {cls}_set_{i["name"]}(const double *input) '
               f'{{ tc_compat_set("{i["name"]}", input); }} \\\n')
    args = ", ".join(f"double *y{k + 1}" for k in range(len(out_meta)))
    cw += (f"void {cls}_getOutputs({args}) {{ double *o_[] = {{"
           + ", ".join(f"y{k + 1}" for k in range(len(out_meta))) + "}; tc_compat_get(o_); }\n")
    w.append("#define TC_COMPAT_WRAPPERS \\\n" + cw)
    w.append('\n#include "tc_runtime.hip.inc"\n')
    return "".join(w)


def class_header(spec):
    """<cls>.h: the reference-exact C symbols (writeCfunctionpergroup.m:513-635)."""
    cls = spec.classname
    s = (f"/* GENERATED: single-instance C surface of class {cls}; same symbols the reference's\n"
         f" * generated {cls}.c exports (writeCfunctionpergroup.m:524,607-635; ipmPD_CSsolver.c:132-141).\n"
         f" * The batched interface is include/tenscalc_hip.h. */\n#include <stdint.h>\n"
         "#ifdef __cplusplus\nextern \"C\" {\n#endif\n")
    for nm, _ in spec.parameters + spec.variables:
        s += f"void {cls}_set_{nm}(const double *input);\n"
    s += (f"void {_solver_of(spec)}(double *mu0, int32_t *maxIter, int32_t *saveIter, double *addEye2Hessian,\n"
          "                    int32_t *status, int32_t *iter, double *time);\n")
    s += f"void {cls}_getOutputs(" + ", ".join(f"double *y{k + 1}" for k in range(len(spec.outputs))) + ");\n"
    s += "#ifdef __cplusplus\n}\n#endif\n"
    return s


def _largest_function(so, prefix="", arch="gfx950"):
    """(name, bytes) of the largest device function of a class library whose name contains `prefix` (code-object symbol
    table; ("", 0) when there is none).  Raises RuntimeError when the LLVM tools that read the code object fail."""
    import re
    import tempfile
    llvm = os.environ.get("TC_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
    with tempfile.TemporaryDirectory() as d:
        fb, co = os.path.join(d, "k.fatbin"), os.path.join(d, "k.co")
        try:
            subprocess.run([os.path.join(llvm, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", so, fb], check=True,
                           capture_output=True)
            subprocess.run([os.path.join(llvm, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fb,
                            "--targets=hipv4-amdgcn-amd-amdhsa--" + arch, "--output=" + co], check=True, capture_output=True)
            txt = subprocess.run([os.path.join(llvm, "llvm-readelf"), "-s", "-W", co], capture_output=True, text=True,
                                 check=True).stdout
        except (OSError, subprocess.CalledProcessError) as e:
            raise RuntimeError("cannot read the code object of %s (%s): %s" % (so, arch, getattr(e, "stderr", None) or e))
    best = ("", 0)
    for ln in txt.splitlines():
        m = re.match(r"\s*\d+:\s+[0-9a-f]+\s+(\d+)\s+FUNC\s+\S+\s+\S+\s+\S+\s+(\S+)", ln)
        if m and prefix in m.group(2) and int(m.group(1)) > best[1]:
            best = (m.group(2), int(m.group(1)))
    return best


def _deps_of(src):
    """the csrc / include files a class source reads, found by following its #include "..." lines (sorted)"""
    import re
    seen, todo = {}, re.findall(r'^\s*#include "([^"]+)"', src, flags=re.M)
    while todo:
        nm = todo.pop()
        if nm in seen:
            continue
        for base in (CSRC, INCLUDE):
            p_ = os.path.join(base, nm)
            if os.path.exists(p_):
                seen[nm] = p_
                todo += re.findall(r'^\s*#include "([^"]+)"', open(p_).read(), flags=re.M)
                break
    return [seen[k] for k in sorted(seen)]


def build_class(spec, fp_contract="fast", arch="gfx950", opt="-O3", block=64, force=False,
                outdir=None, stats=False, tag=None, extra_flags=(), roll=True, min_waves=0, lanes=1,
                recip=False, fast_div=False, inline_max=0, reg_budget=150, scratch="auto", body="code",
                waves=1, fuse=False, static_clean=False, forward=False, rcache=0, lds_state=0, share=False,
                branch_free_tails=True, exact=False, stg_blocks=0, lds_slots=20000, ilp=False, fw_window=0, sched_every=0,
                try_flags=(), refresh_outputs=False, compiled=False, lds_stage=0, key_only=False, lds_split_min=0):
    """Emit <cls>.hip (+ <cls>.h, <cls>.tape) and compile lib<cls>.so with hipcc for gfx950.
    Mirrors the createGateway/compile step of cmex2optimizeCS.m:403-417.

    try_flags: compiler flags worth having but not worth failing for (LLVM's experimental scheduling strategies crash
    the compiler on some classes): tried first, dropped if hipcc fails with them; lib<cls>.flags records what was used.
    A tuple of tuples gives ALTERNATIVES in order of preference (the w8x classes: the experimental iterative-ilp strategy,
    else LLVM's generic `-misched=ilpmax`, else the default).

    key_only: nothing is written or compiled; returns (quick key, source digest) as recorded in <cls><tag>.key /
    <cls><tag>.stamp next to a library -- tests/test_class_digests.py recomputes them from the tracked sources for the
    shipped (git-ignored) libraries.  key_only="quick": the quick key alone, without emitting the source."""
    from .tapeio import save_tape, tape_records, _rec
    cls = spec.classname
    if int(spec.defines.get("verboseLevel", 0)) >= 3:
        # the iteration table is recorded by ONE thread per traced instance (csrc/ipm_newton.hip.inc: TC_V3): the lane-
        # cooperative lowerings evaluate groups with all lanes of an instance and the LDS machine's context has no table
        # pointer; the table shares the handle's diagnostic buffer with the profiling counters
        if body == "ldsm" or lanes != 1:
            raise ValueError("solverVerboseLevel >= 3 (tc_trace) is available on the one-thread-per-instance lowerings only "
                             "(lanes=1, body='code' or 'interp'); got body=%r, lanes=%d" % (body, lanes))
        if stats:
            raise ValueError("solverVerboseLevel >= 3 and stats=%r share the handle's diagnostic buffer: build two classes" % (stats,))
    outdir = outdir or os.path.join(spec.folder, f"@{cls}")
    if not key_only:
        os.makedirs(outdir, exist_ok=True)
    suffix = f"_{tag}" if tag else ""
    # Quick key: the tape, the options and the generator's own sources.  If it matches the one recorded with the
    # library the source need not be emitted again just to be hashed -- lowering mpc_dcmotor for 32 lanes with slot
    # sharing takes two minutes, and every test module asks build() for the whole list of libraries.
    kq = hashlib.sha256()
    for name, arr in tape_records(spec.tape).items():
        kq.update(_rec(name, arr))
    kq.update(repr((cls, sorted(spec.defines.items()), [(n, tuple(sz)) for n, sz in spec.outputs], fp_contract, arch, opt, block,
                    stats, tag, tuple(extra_flags), roll, min_waves, lanes, recip, fast_div, inline_max, reg_budget, scratch,
                    body, waves, fuse, static_clean, forward, rcache, lds_state, share, branch_free_tails, exact, stg_blocks,
                    lds_slots, ilp, fw_window, sched_every, tuple(tuple(a) if isinstance(a, (tuple, list)) else a for a in try_flags), refresh_outputs, compiled, lds_stage, lds_split_min, os.environ.get("TC_LDS_PAD_SLOTS", ""), os.environ.get("TC_STG_BATCH", ""), os.environ.get("TC_LDSM_FUSE", ""), os.environ.get("TC_INTERP_FUSE", ""), os.environ.get("TC_INTERP_PAIRS", ""), os.environ.get("TC_FINE_LIVES", ""), os.environ.get("TC_INTERP_WIDE_MIN", ""), os.environ.get("TC_INTERP_L4_MIN", ""), os.environ.get("TC_LDSM_LOOP_MERGE", ""), os.environ.get("TC_LDSC_DEPTH", ""), os.environ.get("TC_LDSM_PACK_LOCALS", ""), os.environ.get("TC_LDSM_ROW_SKEW", ""), os.environ.get("TC_LDSM_LATE_LAST", ""), os.environ.get("TC_LDSM_PAD_PRODUCTS", ""), os.environ.get("TC_LDSM_MERGE_PRODUCTS", ""), os.environ.get("TC_LDSM_COLD_RANK", ""), os.environ.get("TC_LDSC_TWO_PHASE", ""), os.environ.get("TC_LDSC_SPLIT_HBM", ""), os.environ.get("TC_LDSC_SCHEDB", ""), os.environ.get("TC_LDSC_BYTES", ""), os.environ.get("TC_LDSC_WPE", ""), os.environ.get("TC_LDSM_STAGE", ""), os.environ.get("TC_LDSC_CHUNK", ""), os.environ.get("TC_WPE", ""), os.environ.get("TC_LDSM_SPLIT_MIN", ""), os.environ.get("TC_LDSM_SPLIT_DIV", ""), os.environ.get("TC_LDSM_SPLIT_DIV_MIN", ""), os.environ.get("TC_LDSM_DIV_NEST", ""), os.environ.get("TC_LDSM_PREFIX", ""), os.environ.get("TC_LDSM_MERGE_FOLDS", ""), os.environ.get("TC_INTERP_SPLIT_LONG", ""),
                    os.environ.get("TC_SWEEP_UNROLL", ""))).encode())
    for f_ in sorted(os.listdir(HERE)):
        if f_.endswith(".py"):
            kq.update(open(os.path.join(HERE, f_), "rb").read())
    for d_ in (CSRC, INCLUDE):
        for f_ in sorted(os.listdir(d_)):
            if f_.endswith((".inc", ".h")):
                kq.update(open(os.path.join(d_, f_), "rb").read())
    quick = kq.hexdigest()
    if key_only == "quick":
        return quick
    so_ = os.path.join(outdir, f"lib{cls}{suffix}.so")
    qfile = os.path.join(outdir, f"{cls}{suffix}.key")
    if not key_only and not force and os.path.exists(so_) and os.path.exists(qfile) and open(qfile).read().strip() == quick:
        return so_
    if body == "ldsm":
        if fp_contract != "off":
            raise ValueError("body='ldsm' is a bit-exact back-end: fp_contract='off'")
        src, L = emit_class_ldsm(spec, waves=waves, stats=stats, lds_slots=lds_slots,      # TC_LDSM_FUSE=0: no fused walk
                                 compiled=bool(compiled), lds_stage=lds_stage, min_waves=min_waves, lds_split_min=lds_split_min)
    else:
      src, L = emit_class(spec, fp_contract=fp_contract, block=block, stats=stats, roll=roll,
                        min_waves=min_waves, lanes=lanes, recip=recip, fast_div=fast_div, inline_max=inline_max,
                        reg_budget=reg_budget, scratch=scratch, body=body, waves=waves, fuse=fuse,
                        static_clean=static_clean, forward=forward, rcache=rcache, lds_state=lds_state,
                        share=share, branch_free_tails=branch_free_tails, exact=exact, stg_blocks=stg_blocks, ilp=ilp,
                        fw_window=fw_window, sched_every=sched_every, refresh_outputs=refresh_outputs)
    hip = os.path.join(outdir, f"{cls}{suffix}.hip")
    so = os.path.join(outdir, f"lib{cls}{suffix}.so")
    flags = [opt, f"--offload-arch={arch}", "-fPIC", "-shared", "-std=c++17",
             f"-ffp-contract={fp_contract}", "-I", CSRC, "-I", INCLUDE, "-Wno-unused-value"] + list(extra_flags)
    # (the digest must not depend on where the tree lies: the libraries built here travel to the GPU box, whose
    # checkout is at another path -- with the include directories hashed in, every library was rebuilt there)
    # try_flags: one set of optional flags, or several ALTERNATIVES in order of preference (a tuple of tuples)
    alternatives = [tuple(a) for a in try_flags] if (try_flags and isinstance(try_flags[0], (tuple, list))) else ([tuple(try_flags)] if try_flags else [])
    flat_try = [f for a in alternatives for f in a]
    h = hashlib.sha256((src + " ".join(f for f in flags + flat_try if f not in (CSRC, INCLUDE))).encode())
    for d in _deps_of(src):          # the files this class includes, directly or not: an edit rebuilds only its readers
        h.update(open(d, "rb").read())
    stamp = os.path.join(outdir, f"{cls}{suffix}.stamp")
    digest = h.hexdigest()
    if key_only:
        return quick, digest
    if (not force and os.path.exists(so) and os.path.exists(stamp)
            and open(stamp).read().strip() == digest):
        with open(qfile, "w") as f:
            f.write(quick)
        return so
    with open(hip, "w") as f:
        f.write(src)
    with open(os.path.join(outdir, f"{cls}.h"), "w") as f:
        f.write(class_header(spec))
    from .matlab_class import write_matlab_class
    write_matlab_class(spec, libname=os.path.basename(so), outdir=outdir)     # @<cls>/<cls>.m
    save_tape(spec.tape, os.path.join(outdir, f"{cls}.tape"))
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    r = None
    for used in [list(a) for a in alternatives] + [[]]:          # each alternative in turn, then without any
        cmd = [hipcc] + flags + used + [hip, "-o", so]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if not r.returncode:
            break
    if r.returncode:
        raise RuntimeError(f"hipcc failed for class {cls}:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
    if compiled:
        # The compiled walk's functions must stay below the reach of a 16-bit branch offset (32 767 dwords = 128 KB):
        # beyond it LLVM relaxes branches into s_getpc / s_add / s_setpc sequences on a scavenged SGPR pair -- the one
        # difference between the 384-record build that took a GPU memory-access fault in round 4 and the 96-record builds
        # that never did (scripts/ldsc_chunk_probe.py: 36 relaxed branches in a 35 k-instruction function against none).
        # The cut is by records (chunk = 96: 55 KB for mpc_quadcopter); THIS is the condition it stands for.
        # (Relaxed branches are not a defect by themselves: the tile machine's dispatcher tc_interp is 440 KB with 83 of
        # them and passes every GPU parity test.  The check keeps the compiled walk inside the regime that has been
        # exercised; it is a conservative condition, not a proof of what faulted.)
        try:
            big = _largest_function(so, "tcc_r", arch)
        except RuntimeError:
            os.remove(so)                  # no library without its check (and none without its .stamp / .key / .flags)
            raise
        if big[1] >= 120 * 1024:
            os.remove(so)
            raise RuntimeError(f"compiled LDS walk of {cls}: function {big[0]} is {big[1]} bytes -- beyond the reach of a "
                               "16-bit branch (128 KB); lower the records per function (hip_emitter._ldsc_functions chunk)")
    with open(os.path.join(outdir, f"lib{cls}{suffix}.flags"), "w") as f:
        f.write(" ".join(used) + "\n")
    with open(stamp, "w") as f:
        f.write(digest)
    with open(qfile, "w") as f:
        f.write(quick)
    return so
