"""LDS-resident tape machine (body="ldsm"): storage map and record stream.

One problem instance per workgroup; its hot state lives in the CU's LDS for the whole solve and the lanes of
the workgroup execute DIFFERENT scalar instructions of the same instance (csrc/tc_ldsm.hip.inc).  This is
the back-end for few, large instances -- mpc_quadcopter: 1406x1406 KKT, 67 k instructions, nnz(L) 8 668 --
where one thread per instance (body="interp") leaves the chip empty, pays an HBM round trip per level of
the factorisation and lets the slowest of 64 instances hold a whole wavefront.  The elimination order, the
levels of the symbolic LDL (/root/reference/lib/@csparse/sparsity_ldl.m:161-222) and of the triangular solves
(sparsity_mldivide_l1.m:73-101, _u1.m:185-217) are fixed here, at code-generation time.

Storage.  A scalar instruction's value needs a slot only while something may still read it:
  * internal sets (primal / dual variables, mu, step lengths) and constants: a slot of their own;
  * values of a dependency group that another group or an entry point reads ("exported"): one block per
    group, packed by life time over one pass of the Newton loop (Lowering._assign_shared_slots, the analysis
    the printed-code back-ends use for their LDS slots, at group granularity);
  * values only their own group reads ("local": the L*D products of the factorisation, partial terms): dead as
    soon as the group has been evaluated -- all groups share ONE scratch region.
The whole state of the large problems (25-40 k doubles) does not fit the 160 KB of a CU, so addresses are split:
[0, nLds) is LDS, [nLds, nLds+nGlob) a per-instance array in HBM whose first entries are the instance's inputs.
Blocks are ranked by how much DEEP code (groups with many dependent levels: the factorisation, the triangular
solves) touches them per slot; the best ranked fill the LDS, the rest -- written and read by wide, shallow
levels whose loads overlap -- stay in HBM.

Stream.  Per dependency group the instructions are level-scheduled (as interp_schedule does); the instructions
of one level are cut into RECORDS of up to 64 (one per lane) of the same shape.  A record is a header
(8 words: opcode | kind<<8, n, aux, count, body offset, flags, lane-array length, 0) in a contiguous header
table -- walked with scalar loads -- plus a body of lane arrays in the code array: dest[cnt], operand_j[cnt] for
j < n (kind SHORT, n <= 8) or dest[cnt], len[cnt], head0[cnt], head1[cnt], word_k[cnt] for k < max len rounded up
to 8 (kind LONG: every lane walks its own operand list, strictly left to right; head0/1 = the leading non-product
operands of the dot opcodes).  Operand words are address*2+neg (neg: the operand's sign in I_sum).
"""
from __future__ import annotations

import os

import numpy as np

from . import instructions as I

LANES = 64
NSHORT = 8            # operands of a SHORT record
K_SHORT, K_LONG, K_BARRIER, K_FOLD = 1, 0, 2, 3
NLATE = 2              # late terms a FOLD record can compute itself (body rows 5..13: count, 2 x (position, a, b, c):
                       # the term is a * (b * c), c = the slot holding 1.0 for a plain product)
OP_COPY = 60
OP_CLPTERM = 61       # (b < 0) ? -a/b : DBL_MAX  -- one term of I_clp
OP_PROD3 = 62         # a * (b * c): a term of a dot one of whose factors is itself a product of the level before
OP_PRODDN = 65        # (a / c) * ((b / c) * c): the same term when a and the product's first factor are the quotients N1 / D, N2 / D
                      # of the level before (_div_nest): evaluated from N1, N2, D, one level earlier
DN_SHIFT = 21          # aux bits 21-22 of a dot FOLD record: late term 0 / 1 is of the OP_PRODDN form (uniform over the record)
F_MININF, F_MAXINF = 63, 64   # folds over one chunk of a long minimum / maximum: the chain starts from +inf / -inf
SPLIT_MIN = 96        # a min / max / clp / long sum with more terms is folded in two levels (build_stream; TC_LDSM_SPLIT_MIN overrides)
SPLIT_CHUNK = 32
F_SUM = 3             # fold opcodes of K_FOLD records (the dot opcodes and I_sumprod keep their own number)
STAGE = 2048          # staging slots for the terms of long instructions
F_GLOBAL_SRC, F_GLOBAL_DST, F_NEW_LEVEL = 1, 2, 4
F_NLATE_SHIFT = 3      # flags bits 3-4 of a FOLD record: the largest late-term count among its lanes
F_PADDED = 32          # FOLD record: every row is padded to a multiple of 8 terms with the operation's identity
F_BARRIER_AFTER = 64   # the step ends behind this record: workgroup barrier (W > 1), saves a barrier record
LATE_LAST = 1 << 20    # aux bit of a dot FOLD record: the late terms are the LAST terms of every row -- they are not staged;
                       # the lane folds the (padded) early terms and adds the late products behind them.  x + (-0.0) is x,
                       # so "early, pads, late" is "early, late, pads": the reference's chain
F_BARRIER_GLOBAL = 128  # ... and the step wrote to the instance's HBM array: the barrier must order those stores too
_DOT_SKIP = {I.I_minus_dot: 0, I.I_minus_dot_div: 1, I.I_plus_minus_dot: 1, I.I_plus_minus_dot_div: 2}
_LONG_OPS = {I.I_sum, I.I_sumprod, I.I_plus_sqr, I.I_plus_abs, I.I_min, I.I_min0, I.I_max, I.I_max0,
             I.I_max_abs, I.I_clp, I.I_minus_dot, I.I_minus_dot_div, I.I_plus_minus_dot, I.I_plus_minus_dot_div}
DEEP = 16             # a group with at least this many dependent levels is latency critical


def group_levels(T):
    """-> (level of every instruction inside its group [n+1], number of levels per group [nGroups+1])"""
    opl, ost = T.ops.tolist(), T.opStart.tolist()
    grp = T.group.tolist()
    lvl = [0] * (T.n + 1)
    nlev = [0] * (T.nGroups + 1)
    for g, members in enumerate(T.groups, 1):
        mx = -1
        for i in members.tolist():
            l_ = 0
            for j in opl[ost[i - 1]:ost[i]]:
                if grp[j] == g and lvl[j] + 1 > l_:
                    l_ = lvl[j] + 1
            lvl[i] = l_
            if l_ > mx:
                mx = l_
        nlev[g] = mx + 1
    return lvl, nlev


_DOTS = (I.I_minus_dot, I.I_minus_dot_div, I.I_plus_minus_dot, I.I_plus_minus_dot_div)


def _is_product(T, opl, ost, j):
    """instruction j is v = p * q (I_sumprod, one term of two factors)"""
    return int(T.types[j - 1]) == I.I_sumprod and ost[j] - ost[j - 1] == 2 and int(T.pars[T.parStart[j - 1]]) == 2


def _nested_factor(T, opl, ost, g, ids):
    """which factor (0 / 1) of the term ids=(a, b) of a dot may be expanded into its own two factors: the second if
    it is a plain product of group g, else the first; never both (stream_levels and build_stream must agree)"""
    for k in (1, 0):
        j = ids[k]
        if T.group[j] == g and T.used[j] and _is_product(T, opl, ost, j):
            return k
    return None


class _DN(list):
    """operand words [N1, N2, D] of a term in the OP_PRODDN form (build_stream)"""


def _div_nest(T, opl, ost, g, ids, k):
    """The term a * v of a dot, v = p * q the plain product that _nested_factor chose (factor k), is a term of a symbolic
    LDL' written with its divisions spelled out (split_div_tape): a = N1 / D and one factor of v is N2 / D with the other
    factor D itself -- L(r,i) * (L(c,i) * D(i)) = (N1/D) * ((N2/D) * D).  -> (N1, N2, D), the three values the term can be
    evaluated from ONE LEVEL EARLIER than from a, p, q (the divisions are done by the lane that needs the term: the same
    roundings in the same order, a / D and p / D rounded as the I_div instructions round them), or None.
    OFF unless TC_LDSM_DIV_NEST=1: measured on mpc_quadcopter (profiles/r06_quadcopter_div_nest.log) the walk has 12 % fewer
    steps and is 8 % SLOWER -- two correctly rounded divisions on the critical path of every fold step cost more than the
    steps they remove; bit-identical either way."""
    if os.environ.get("TC_LDSM_DIV_NEST", "0") != "1":
        return None
    a, v = ids[1 - k], ids[k]
    ty = T.types
    if int(ty[a - 1]) != I.I_div or T.group[a] != g or not T.used[a]:
        return None
    n1, d = opl[ost[a - 1]:ost[a]]
    pj = opl[ost[v - 1]:ost[v]]
    for p_, q_ in ((pj[0], pj[1]), (pj[1], pj[0])):
        if q_ == d and int(ty[p_ - 1]) == I.I_div and T.group[p_] == g and T.used[p_]:
            n2, d2 = opl[ost[p_ - 1]:ost[p_]]
            if d2 == d:
                return n1, n2, d, (pj[0] == p_)          # last: the division is the FIRST factor of v = p * q (operand order)
    return None


def _dot_level(T, opl, ost, grp, lvl, g, ops, skip, ignore=None):
    """level of a staged dot of group g with operands `ops` (the first `skip` are heads) from the levels of what it reads:
    a factor that is a plain product of the group counts through ITS factors (nested product), a term of the form
    (N1/D) * ((N2/D) * D) through N1, N2 and D (_div_nest).  ignore: an operand position left out (split_div_tape: the divisor)"""
    l_ = 0
    expand = set()
    divn = {}
    for q in range(skip, len(ops) - 1, 2):
        k = _nested_factor(T, opl, ost, g, ops[q:q + 2])
        if k is not None:
            expand.add(q + k)
            dn = _div_nest(T, opl, ost, g, ops[q:q + 2], k)
            if dn is not None:                  # both operands of the pair are replaced by N1, N2, D
                divn[q] = divn[q + 1] = dn[:3]
    for q, j in enumerate(ops):
        if q == ignore:
            continue
        if q in divn:
            for jj in divn[q]:
                if grp[jj] == g and lvl[jj] + 1 > l_:
                    l_ = lvl[jj] + 1
            continue
        if grp[j] != g:
            continue
        if q in expand:
            for jj in opl[ost[j - 1]:ost[j]]:
                if grp[jj] == g and lvl[jj] + 1 > l_:
                    l_ = lvl[jj] + 1
        elif lvl[j] + 1 > l_:
            l_ = lvl[j] + 1
    return l_


def stream_levels(T, nested=True):
    """Levels for the record stream.  As group_levels, except that a staged dot product (more than NSHORT operands)
    does not wait for a factor that is itself a plain product v = p*q of its group: the term a*v can be evaluated
    as the nested product a*(p*q) -- the same two roundings in the same order -- from operands that are one level
    older.  In the symbolic LDL (sparsity_ldl.m:161-222) v_i = L(c,i)*D(i) is such a product: D(c) and L(r,c) move
    up a level, two levels per column instead of three."""
    opl, ost = T.ops.tolist(), T.opStart.tolist()
    grp = T.group.tolist()
    lvl = [0] * (T.n + 1)
    nlev = [0] * (T.nGroups + 1)
    for g, members in enumerate(T.groups, 1):
        mx = -1
        for i in members.tolist():
            ops = opl[ost[i - 1]:ost[i]]
            op = int(T.types[i - 1])
            dot = nested and op in _DOTS and len(ops) > NSHORT
            skip = _DOT_SKIP.get(op, 0) if dot else 0
            if dot:
                l_ = _dot_level(T, opl, ost, grp, lvl, g, ops, skip)
            else:
                l_ = 0
                for j in ops:
                    if grp[j] == g and lvl[j] + 1 > l_:
                        l_ = lvl[j] + 1
            lvl[i] = l_
            if l_ > mx:
                mx = l_
        nlev[g] = mx + 1
    return lvl, nlev


def fusion_sets(T, min_gain=32):
    """Sets of dependency groups worth walking as ONE level schedule: a deep group B (forward / backward substitution)
    that reads a deep group A (the KKT factorisation) waits for the LAST level of A although its row r only needs row r
    of the factor -- scheduled together, the forward substitution rides on the factorisation's levels (same
    instructions, same operand order: only the step an instruction is assigned to changes).  Groups on a dependency
    path from A to B join the set (a merge must not close a cycle, csparse.Tape.merged).  Greedy by the number of
    levels saved; at least `min_gain`.  mpc_quadcopter: {factor, 28 instructions in between, solves}: 507 + 13 + 523
    levels -> 776."""
    lvl, nlev = stream_levels(T)
    opl, ost = T.ops.tolist(), T.opStart.tolist()
    grp = T.group
    nG = T.nGroups
    preds = [set() for _ in range(nG + 1)]
    for i in range(1, T.n + 1):
        gi = int(grp[i])
        if gi and T.used[i]:
            for j in opl[ost[i - 1]:ost[i]]:
                gj = int(grp[j])
                if gj and gj != gi:
                    preds[gi].add(gj)
    anc = [set() for _ in range(nG + 1)]            # transitive predecessors (groups are topologically numbered)
    for g in range(1, nG + 1):
        for p_ in preds[g]:
            anc[g] |= anc[p_] | {p_}
    deep = [g for g in range(1, nG + 1) if nlev[g] >= DEEP]
    cands = []
    for b in deep:
        for a in deep:
            if a in preds[b]:
                between = {c for c in anc[b] if a in anc[c]}
                cands.append((a, b, sorted({a, b} | between)))
    out, taken = [], set()
    scored = []
    for a, b, S in cands:
        if len(S) > 8:
            continue
        T2 = T.merged([S])
        _, nl2 = stream_levels(T2)
        gain = sum(nlev[g] for g in S) - max(nl2)        # the merged group is the deepest of T2 or close to it
        g2 = int(T2.group[int(T.groups[a - 1][0])])
        gain = sum(nlev[g] for g in S) - nl2[g2]
        scored.append((gain, S))
    for gain, S in sorted(scored, key=lambda x: -x[0]):
        if gain >= min_gain and not (set(S) & taken):
            out.append(S)
            taken |= set(S)
    return out


def loop_classes(T, defines):
    """Sets of dependency groups that the Newton loop evaluates at exactly the same call sites and invalidates with the
    same entry points (loop_schedule.timeline: positions where a group may be evaluated; parameter sets are not part of a
    solve and do not count).  The reference keeps them apart because some OTHER program could call their entry points
    separately; inside one solve kernel they are one group -- one evaluation, one level schedule in which independent
    members overlap, no hand-over between them (a group transition costs the compiled walk ~1 300 cycles: return, flag
    test, workgroup barrier, dispatch, the first records' body words).  A set must be convex (no outside group on a path
    between two members) and the sets together must leave the group graph acyclic; members are dropped until they do."""
    from .loop_schedule import timeline
    try:
        E = timeline(T, defines)["E"]
    except Exception:           # a loop this analysis does not know (equilibrium solvers): no merging
        return []
    nG = T.nGroups
    opl, ost = T.ops.tolist(), T.opStart.tolist()
    grp = T.group
    preds = [set() for _ in range(nG + 1)]
    for i in range(1, T.n + 1):
        gi = int(grp[i])
        if gi and T.used[i]:
            for j in opl[ost[i - 1]:ost[i]]:
                gj = int(grp[j])
                if gj and gj != gi:
                    preds[gi].add(gj)
    anc = [set() for _ in range(nG + 1)]
    for g in range(1, nG + 1):
        for p_ in preds[g]:
            anc[g] |= anc[p_] | {p_}
    ext = {e["name"] for e in T.sets if e["external"]}
    inval = [set() for _ in range(nG + 1)]
    for e in list(T.sets) + list(T.copies):
        if e["name"] not in ext:
            for c_ in e["children"]:
                inval[c_].add(e["name"])
    # ... and needed by the same entry points: the initialisation sequence is ONE position of the timeline, but a group
    # that scaleIneq__ needs must not be evaluated with one that initDualEqX__ needs -- its inputs are not there yet
    # (the CPU model of tests/test_ldsm_stream.py caught exactly that on mpc_dcmotor; the tile machine, whose values have rows
    # of their own, was tried on classes by position alone as well: l1l2 no longer bit-identical)
    needed = [set() for _ in range(nG + 1)]
    for e in list(T.gets) + list(T.copies):
        for p_ in e["parents"]:
            needed[p_].add(e["name"])
    classes = {}
    for g in range(1, nG + 1):
        if E.get(g):
            classes.setdefault((tuple(E[g]), tuple(sorted(needed[g])), tuple(sorted(inval[g]))), []).append(g)
    sets = []
    for members in classes.values():
        # convex subsets, greedily in topological (= numerical) order
        rest = list(members)
        while len(rest) > 1:
            cur = [rest[0]]
            for g in rest[1:]:
                between = {x for x in anc[g] if any(c_ in anc[x] for c_ in cur)} - set(cur)
                if not between:
                    cur.append(g)
            if len(cur) > 1:
                sets.append(cur)
            rest = [g for g in rest if g not in cur]
    # jointly acyclic: contract, Kahn; a set on a cycle loses its last member until the graph is a DAG
    while True:
        rep = list(range(nG + 1))
        for S in sets:
            for g in S:
                rep[g] = S[0]
        succ, indeg = {}, {}
        nodes = {rep[g] for g in range(1, nG + 1)}
        for g in range(1, nG + 1):
            for p_ in preds[g]:
                a, b = rep[p_], rep[g]
                if a != b and b not in succ.setdefault(a, set()):
                    succ[a].add(b)
                    indeg[b] = indeg.get(b, 0) + 1
        ready = [x for x in nodes if not indeg.get(x)]
        seen = 0
        while ready:
            x = ready.pop()
            seen += 1
            for y in succ.get(x, ()):
                indeg[y] -= 1
                if not indeg[y]:
                    ready.append(y)
        if seen == len(nodes):
            break
        stuck = {x for x in nodes if indeg.get(x)}
        cand = [S for S in sets if S[0] in stuck]
        if not cand:
            raise AssertionError("cycle among unmerged groups")
        S = min(cand, key=len)
        S.pop()
        sets = [S_ for S_ in sets if len(S_) > 1]
    return sets


def split_div_tape(T, min_operands=NSHORT + 1):
    """The tape with the long `(h0 - sum) / h1` (I_plus_minus_dot_div) and `(-sum) / h0` (I_minus_dot_div) whose DIVISOR is
    what they wait for written as TWO instructions, the numerator (I_plus_minus_dot / I_minus_dot: the same chain, rounded
    the same) and an I_div -- bit for bit the same values: the reference's statement rounds the numerator before it
    divides (instructionsTableUTHash.c:1067-1108).

    Why: in the symbolic LDL' (sparsity_ldl.m:161-222) L(r,c) = (A(r,c) - sum_i L(r,i) v_i) / D(c) waits for D(c), which is
    itself a fold of column c's row that ends in the same late term -- two dependent fold steps per column.  The
    numerator needs nothing that D(c) does not need as well: split, it is folded IN THE STEP of D(c) (other lanes, other
    wavefronts), and the second step of the column is one short division record instead of a second row read, chain and
    late term.  An instruction whose numerator is as late as its divisor is left alone (the split would cost it a
    level).  Same dependency groups (T's own, merged ones included), entry points remapped; the new values are
    group-local."""
    n = T.n
    ost, opl = T.opStart.tolist(), T.ops.tolist()
    pst, prl = T.parStart.tolist(), T.pars.tolist()
    ty = T.types.tolist()
    grp = T.group.tolist()
    lvl, _ = stream_levels(T, True)

    def numerator_level(i, g, ops, skip, div_pos):
        return _dot_level(T, opl, ost, grp, lvl, g, ops, skip, ignore=div_pos)
    new = [0] * (n + 1)
    types2, ops2, os2, pars2, ps2 = [], [], [0], [], [0]
    extra = {}                                    # group -> new ids of the numerators

    def put(t, o, p):
        types2.append(t)
        ops2.extend(o)
        os2.append(len(ops2))
        pars2.extend(p)
        ps2.append(len(pars2))
        return len(types2)
    for i in range(1, n + 1):
        t = ty[i - 1]
        old_ops = opl[ost[i - 1]:ost[i]]
        o = [new[j] for j in old_ops]
        p = prl[pst[i - 1]:pst[i]]
        # (the numerator must stay a staged dot -- more than NSHORT operands -- or it loses its nested late terms)
        if T.used[i] and grp[i] and len(o) >= max(min_operands, NSHORT + 2) and t in (I.I_plus_minus_dot_div, I.I_minus_dot_div):
            dpos = 1 if t == I.I_plus_minus_dot_div else 0
            if numerator_level(i, grp[i], old_ops, _DOT_SKIP[t], dpos) < lvl[i]:
                if t == I.I_plus_minus_dot_div:
                    num = put(I.I_plus_minus_dot, [o[0]] + o[2:], p)
                else:
                    num = put(I.I_minus_dot, o[1:], p)
                new[i] = put(I.I_div, [num, o[dpos]], [])
                extra.setdefault(grp[i], []).append(num)
                continue
        new[i] = put(t, o, p)
    if not extra:
        return T
    rm = np.asarray(new, dtype=np.int64)
    sets = [dict(name=e["name"], varname=e["varname"], size=e["size"], ins=rm[e["ins"]], external=e["external"]) for e in T.sets]
    gets = [dict(name=e["name"], outputs=[dict(name=o_["name"], size=o_["size"], ins=rm[o_["ins"]]) for o_ in e["outputs"]])
            for e in T.gets]
    copies = [dict(name=e["name"], dst=rm[e["dst"]], src=rm[e["src"]]) for e in T.copies]
    merge = [np.concatenate([rm[members], np.asarray(extra.get(g, []), dtype=np.int64)]) for g, members in enumerate(T.groups, 1)]
    from .csparse import Tape
    return Tape.from_instructions(np.asarray(types2, dtype=np.int32), np.asarray(os2, dtype=np.int64),
                                  np.asarray(ops2, dtype=np.int64), np.asarray(ps2, dtype=np.int64),
                                  np.asarray(pars2, dtype=np.float64), sets, gets, copies, T.defines, ldl_stats=T.ldl_stats,
                                  merge=merge)


def fused_tape(T, defines, split_div=False):
    """the tape the LDS machine walks: the reference's instructions and entry points, coarser dependency groups --
    factorisation + substitutions as one level schedule (fusion_sets), then the groups one solve cannot tell apart
    (loop_classes).  Values are unchanged.  TC_LDSM_FUSE=0 / TC_LDSM_LOOP_MERGE=0 switch the steps off.
    split_div: split_div_tape afterwards (hip_emitter.emit_class_ldsm asks for it when TC_LDSM_SPLIT_DIV=1; off by default:
    no gain by itself -- a step with one short division record costs what a fold step costs -- and slower with _div_nest)."""
    if os.environ.get("TC_LDSM_FUSE", "1") != "0":
        fs = fusion_sets(T)
        if fs:
            T = T.merged(fs)
        if os.environ.get("TC_LDSM_LOOP_MERGE", "1") != "0":
            cs = loop_classes(T, defines)
            if cs:
                # (Tape.merged starts from the reference's partition: every group of THIS tape is passed, the ones that
                # stay alone as sets of one, or the first step's merge would be undone)
                taken = {g for S in cs for g in S}
                T = T.merged(cs + [[g] for g in range(1, T.nGroups + 1) if g not in taken])
    if split_div:
        # (levels are those of the merged groups.  Repeated: once the divisions of a column are instructions of their own,
        # the terms built on them move up a level (_div_nest), and more numerators turn out to be earlier than their divisor)
        for _ in range(8):
            T2 = split_div_tape(T, int(os.environ.get("TC_LDSM_SPLIT_DIV_MIN", NSHORT + 1)))
            if T2 is T:
                break
            T = T2
    return T


def assign_slots(T, defines, lds_slots=20000, stage=STAGE, pack_locals=None):
    """-> dict(addr=int array [n+1] (-1: no storage), nLds, nGlob, nExt, consts=[(addr, value)],
    inputs=[(first global index, numel, entry name)], dummy=LDS slot nobody reads, ...)"""
    from .lowering import Lowering
    stage = int(os.environ.get("TC_LDSM_STAGE", stage))      # staging slots: tapes without long folds need few
    L = Lowering(T, None, roll=False, lanes=1, spill_all=True)
    n = T.n
    grp = T.group
    opl, ost = T.ops.tolist(), T.opStart.tolist()
    lvl, nlev = group_levels(T)
    read_other = np.zeros(n + 1, dtype=bool)
    read_any = np.zeros(n + 1, dtype=bool)
    heat = np.zeros(n + 1)                       # reads by deep groups
    for i in range(1, n + 1):
        if not T.used[i]:
            continue
        gi = int(grp[i])
        deep = gi > 0 and nlev[gi] >= DEEP
        for j in opl[ost[i - 1]:ost[i]]:
            read_any[j] = True
            if grp[j] != gi:
                read_other[j] = True
            if deep:
                heat[j] += 1
        if deep:
            heat[i] += 1                         # written by a deep group
    direct = L.out_needed                      # read by a get / copy source
    # ---- requests -------------------------------------------------------------------------------------------
    where = np.full(n + 1, -1, dtype=np.int64)   # final address
    inputs, nExt = [], 0
    gslot = {}                                   # instruction -> global index
    for e in T.sets:
        if e["external"]:
            ins = [int(i) for i in e["ins"]]
            for q, i in enumerate(ins):
                gslot[i] = nExt + q
            inputs.append((nExt, len(ins), e["name"]))
            nExt += len(ins)
    lds = 0
    dummy = lds
    lds += 1
    zero = lds                                   # always 0.0: padding operand
    lds += 1
    negzero = lds                                # always -0.0: x + (-0.0) is x, pads the rows of ordered sums
    lds += 1
    posinf, neginf = lds, lds + 1                # pads of the rows of minima / maxima
    one = lds + 2                                # third factor of a late term that is a plain product
    lds += 3
    consts = [(zero, 0.0), (negzero, -0.0), (posinf, float("inf")), (neginf, float("-inf")), (one, 1.0)]
    for i in range(1, n + 1):
        if T.isload[i - 1] and (read_any[i] or direct[i]):
            where[i] = lds
            consts.append((lds, float(T.pars[T.parStart[i - 1]])))
            lds += 1
    for i in range(1, n + 1):                    # internal sets / copy destinations: the iterate, mu, alphas
        if T.isset[i - 1] and i not in gslot and (read_any[i] or direct[i]):
            where[i] = lds
            lds += 1
    # exported values: one block per group, packed by life time; locals: one shared scratch region
    blocks, locals_ = [], {}
    for g, members in enumerate(T.groups, 1):
        ex, lo = [], []
        for i in members.tolist():
            if T.used[i]:
                (ex if (read_other[i] or direct[i]) else lo).append(i)
        if ex:
            blocks.append(ex)
        locals_[g] = lo
    # Group-local values are packed by LIFE TIME inside their group (round 4): the walk is level by level, a value is
    # written at its level and read at its readers' levels (a reader that expands a product into its factors,
    # stream_levels, reads the factors at ITS level), so a slot is free for a value of a later level as soon as its
    # occupant's last reader has run -- a barrier (W > 1) or the single wavefront's program order lies in between.  For
    # mpc_quadcopter's factor + solves 11 463 values need 5 914 slots; what this frees takes the group's INPUTS into the
    # LDS: before, 575 of its 800 levels waited for a value in HBM.  (TC_LDSM_PACK_LOCALS=0: one slot per value.)
    if pack_locals is None:
        pack_locals = os.environ.get("TC_LDSM_PACK_LOCALS", "1") != "0"
    local_slot = {}
    sizes = [0]
    local_sizes = {}                             # diagnostics: packed group-local slots per group
    if pack_locals:
        slvl, _ = stream_levels(T, True)
        for g, lo in locals_.items():
            if not lo:
                continue
            los = set(lo)
            last = {i: slvl[i] for i in lo}
            for i in T.groups[g - 1].tolist():
                if not T.used[i]:
                    continue
                li = slvl[i]
                ops_i = opl[ost[i - 1]:ost[i]]
                if int(T.types[i - 1]) in _DOTS and len(ops_i) > NSHORT:      # ... or N1, N2, D of a term (_div_nest)
                    for q in range(_DOT_SKIP[int(T.types[i - 1])], len(ops_i) - 1, 2):
                        k_ = _nested_factor(T, opl, ost, g, ops_i[q:q + 2])
                        dn_ = _div_nest(T, opl, ost, g, ops_i[q:q + 2], k_) if k_ is not None else None
                        for jj in (dn_[:3] if dn_ else ()):
                            if jj in los and li > last[jj]:
                                last[jj] = li
                for j in ops_i:
                    if j in los:
                        if li > last[j]:
                            last[j] = li
                        if _is_product(T, opl, ost, j):          # i may read j's factors instead of j
                            for jj in opl[ost[j - 1]:ost[j]]:
                                if jj in los and li > last[jj]:
                                    last[jj] = li
                    elif grp[j] == g and _is_product(T, opl, ost, j):
                        for jj in opl[ost[j - 1]:ost[j]]:
                            if jj in los and li > last[jj]:
                                last[jj] = li
            import heapq
            free, busy, nxt = [], [], 0       # free slot numbers; (last level, slot) of occupied ones
            for i in sorted(lo, key=lambda i_: (slvl[i_], i_)):
                while busy and busy[0][0] < slvl[i]:
                    heapq.heappush(free, heapq.heappop(busy)[1])
                if free:
                    q = heapq.heappop(free)
                else:
                    q = nxt
                    nxt += 1
                local_slot[i] = q
                heapq.heappush(busy, (last[i], q))
            sizes.append(nxt)
            local_sizes[g] = nxt
        nLocal = max(sizes)
    else:
        nLocal = max([len(v) for v in locals_.values()], default=0)
    local_base = lds
    lds += nLocal
    stage_base = lds                             # terms of long instructions (build_stream), + 8: chunked reads
    lds += int(stage) + 8
    for g, lo in locals_.items():
        for q, i in enumerate(lo):
            where[i] = local_base + (local_slot[i] if pack_locals else q)
    # hot blocks into what is left of the LDS (each packed among themselves by life time), cold blocks to HBM
    L.share = defines
    L.ignore_dead_readers = True
    # ranking: blocks the deep groups touch first, by reads per value (a cold value there stalls a latency-critical level);
    # the others by the number of STEPS that touch them per slot they would take -- a step with one operand in HBM
    # costs as much as one with sixty (round 4: mpc_quadcopter spent 13 % of an iteration in such steps)
    # (weighting the groups the Newton loop never evaluates down to 0.05 was measured: no better than no ranking)
    touch = [set() for _ in blocks]
    if os.environ.get("TC_LDSM_COLD_RANK", "1") != "0":
        bof = np.full(n + 1, -1, dtype=np.int64)
        for b, bl in enumerate(blocks):
            bof[bl] = b
        for i in range(1, n + 1):
            if not T.used[i]:
                continue
            key = (int(grp[i]), int(lvl[i]))
            if bof[i] >= 0:
                touch[bof[i]].add(key)
            for j in opl[ost[i - 1]:ost[i]]:
                if bof[j] >= 0:
                    touch[bof[j]].add(key)
    score = [len(touch[b]) / len(blocks[b]) for b in range(len(blocks))]
    order = sorted(range(len(blocks)), key=lambda b: (-float(np.mean(heat[blocks[b]])), -score[b]))
    hot, cold = [], []
    room = int(lds_slots) - lds
    if room < 0:
        raise ValueError("ldsm: persistent state + local scratch (%d slots) exceed the LDS budget" % lds)

    def packed(bl):
        if not bl:
            return 0, None
        ps = np.full(n + 1, -1, dtype=np.int64)
        return L._assign_shared_slots([blocks[b] for b in bl], ps), ps
    # largest prefix of the ranking that fits (packing is not monotone in general: check each candidate size)
    lo_, hi_ = 0, len(order)
    best = (0, None)
    while lo_ < hi_:
        mid = (lo_ + hi_ + 1) // 2
        sz, ps = packed(order[:mid])
        if sz <= room:
            lo_, best = mid, (sz, ps)
        else:
            hi_ = mid - 1
    hot, cold = order[:lo_], order[lo_:]
    # ... then every later block of the ranking that still fits (merged groups export large blocks: the first one that
    # does not fit used to end the list with 2 000 slots to spare)
    if os.environ.get("TC_LDSM_COLD_RANK", "1") != "0":
        for b in list(cold):
            sz, ps = packed(hot + [b])
            if sz <= room:
                hot.append(b)
                cold.remove(b)
                best = (sz, ps)
    nHot, ps = best
    for b in hot:
        for i in blocks[b]:
            where[i] = lds + ps[i]
    lds += nHot
    nLds = lds
    nCold, ps = packed(cold)
    for b in cold:
        for i in blocks[b]:
            where[i] = nLds + nExt + ps[i]
    for i, q in gslot.items():
        where[i] = nLds + q
    return dict(addr=where, nLds=nLds, nGlob=nExt + nCold, nExt=nExt, consts=consts, inputs=inputs, dummy=dummy,
                zero=zero, negzero=negzero, posinf=posinf, neginf=neginf, one=one, stage_base=stage_base, nStage=int(stage), nLocal=nLocal,
                local_sizes=local_sizes, nHot=nHot, nCold=nCold, hot_blocks=len(hot), cold_blocks=len(cold),
                lowering=L, levels=lvl, nlev=nlev)


# ---------------------------------------------------------------------------------------------------------------
def _rec(words, headers, op, kind, n, aux, items, nLds, new_level):
    """one record: items = [(dest addr, [operand words])] (<= 64, same shape for SHORT).

    Body layout.  SHORT and FOLD: array of structures -- lane q's words are contiguous at off + q*S, S = the row
    count rounded up to a multiple of 4, so that a lane fetches its body with one to three 16-byte loads
    (SHORT: dest, operands; FOLD: dest, len, head0, head1, base and, with late terms, nlate | pos0 << 2 | pos1 << 17,
    a0, b0, c0, a1, b1, c1).  LONG: structure of arrays, rows of cn words (dest, len, head0, head1, then the operand
    lists, which a lane walks chunk by chunk)."""
    cnt = len(items)
    cn = -(-cnt // 8) * 8
    flags = F_NEW_LEVEL if new_level else 0
    if any((w >> 1) >= nLds for _, a in items for w in a):
        flags |= F_GLOBAL_SRC
    if any(d >= nLds for d, _ in items):
        flags |= F_GLOBAL_DST
    off = len(words)
    if off % 4:
        raise AssertionError("record bodies start on 16-byte boundaries")
    pad = lambda v, fill: list(v) + [fill] * (cn - len(v))        # noqa: E731

    def aos(rows):
        S = -(-len(rows[0]) // 4) * 4
        for q in range(cn):
            r = rows[min(q, cnt - 1)]
            words.extend(list(r) + [0] * (S - len(r)))
    if kind == K_SHORT:
        aos([[d] + list(a[:n]) for d, a in items])
        nn = n
    elif kind == K_FOLD:      # items: (dest, [len, head0, head1, base (, nlate, NLATE x (position, a, b, c))])
        rows = len(items[0][1])
        if any(len(a) != rows for _, a in items):
            raise AssertionError("fold record: mixed body shapes")
        if rows > 4 and max(a[4] for _, a in items) == 0:       # no lane of this record has a late term
            items = [(d, a[:4]) for d, a in items]
            rows = 4
        if rows > 4:
            if NLATE != 2 or any(a[5] >= 1 << 15 or a[9] >= 1 << 14 for _, a in items):
                raise AssertionError("late-term positions do not fit the packed word")
            aos([[d] + list(a[:4]) + [a[4] | a[5] << 2 | a[9] << 17, a[6], a[7], a[8], a[10], a[11], a[12]] for d, a in items])
        else:
            aos([[d] + list(a) for d, a in items])
        nn = -(-max(a[0] for _, a in items) // 8) * 8
        if op in _PADDED_FOPS:
            flags |= F_PADDED
        flags &= ~F_GLOBAL_SRC
        if any((a[j] >> 1) >= nLds for _, a in items for j in (1, 2)):
            flags |= F_GLOBAL_SRC
        if rows > 4:
            flags |= max(a[4] for _, a in items) << F_NLATE_SHIFT
            if any((a[j] >> 1) >= nLds for _, a in items for q in range(NLATE) for j in (6 + 4 * q, 7 + 4 * q, 8 + 4 * q)):
                flags |= F_GLOBAL_SRC
    else:
        words += pad([d for d, _ in items], items[-1][0])
        skip = _DOT_SKIP.get(op, 0)
        lens = [len(a) - skip for _, a in items]
        mx = -(-max(lens) // 8) * 8
        words += pad(lens, 0)
        for q in range(2):                                  # heads of the dot opcodes (zero slot when unused)
            words += pad([a[q] if q < skip else 2 for _, a in items], 2)
        for k in range(mx):
            words += pad([a[skip + k] if skip + k < len(a) else 2 for _, a in items], 2)     # 2 = the zero slot (addr 1)
        nn = mx
    headers += [op | (kind << 8), nn, aux, cnt, off, flags, cn, 0]


def record_body(words, hdr):
    """logical body rows of a record, [row][lane < cnt]: SHORT dest, operands; FOLD dest, len, head0, head1, base
    (, nlate, pos0, a0, b0, c0, pos1, a1, b1, c1); LONG dest, len, head0, head1, operand words"""
    w0, n, aux, cnt, off, flags, cn, _ = (int(x) for x in hdr)
    kind = w0 >> 8
    if kind == K_LONG:
        na = 4 + n
        if off + na * cn > len(words):
            raise AssertionError("record body outside the code array")
        return np.asarray(words[off:off + na * cn], dtype=np.int64).reshape(na, cn)[:, :cnt]
    nlm = (flags >> F_NLATE_SHIFT) & 3 if kind == K_FOLD else 0
    na = 1 + n if kind == K_SHORT else (12 if nlm else 5)
    S = -(-na // 4) * 4
    if off + S * cn > len(words):
        raise AssertionError("record body outside the code array")
    b = np.asarray(words[off:off + S * cn], dtype=np.int64).reshape(cn, S).T[:na, :cnt]
    if nlm:
        pk = b[5]
        b = np.vstack([b[:5], pk & 3, (pk >> 2) & 0x7fff, b[6:9], pk >> 17, b[9:12]])
    return b


# the rows of every fold are padded to a multiple of 8 terms with the operation's identity (index into `pads`:
# 0 = -0.0, 1 = +inf, 2 = -inf): the chain then needs neither a select nor a clamped address per term
_PADDED_FOPS = {I.I_minus_dot: 0, I.I_minus_dot_div: 0, I.I_plus_minus_dot: 0, I.I_plus_minus_dot_div: 0, F_SUM: 0,
                I.I_sumprod: 0, I.I_min: 1, I.I_min0: 1, I.I_clp: 1, I.I_max: 2, I.I_max0: 2, F_MININF: 1, F_MAXINF: 2}
# two-level folds: fold opcode of the instruction -> (fold opcode of a chunk, the first term leads the second level)
_SPLIT = {I.I_sumprod: (F_SUM, False), I.I_min: (F_MININF, True), I.I_min0: (F_MININF, False), I.I_clp: (F_MININF, False),
          I.I_max: (F_MAXINF, True), I.I_max0: (F_MAXINF, False)}
_TERM = {  # long instruction -> (term opcode, operands per term, fold opcode); terms are evaluated in parallel, one per
    # lane, into consecutive staging slots; ONE lane then folds them strictly left to right
    I.I_minus_dot: ("prod", 2, I.I_minus_dot), I.I_minus_dot_div: ("prod", 2, I.I_minus_dot_div),
    I.I_plus_minus_dot: ("prod", 2, I.I_plus_minus_dot), I.I_plus_minus_dot_div: ("prod", 2, I.I_plus_minus_dot_div),
    I.I_plus_sqr: (I.I_sqr, 1, F_SUM), I.I_plus_abs: (I.I_abs, 1, F_SUM), I.I_max_abs: (I.I_abs, 1, I.I_max),
    I.I_min: (OP_COPY, 1, I.I_min), I.I_min0: (OP_COPY, 1, I.I_min0), I.I_max: (OP_COPY, 1, I.I_max),
    I.I_max0: (OP_COPY, 1, I.I_max0), I.I_sum: (I.I_sum, 1, F_SUM), I.I_clp: (OP_CLPTERM, 2, I.I_clp),
}


def _split_chunk(fop, aux, nterm, split=True):
    """chunk length of a two-level fold, or 0"""
    # split: False (never), True (more than SPLIT_MIN terms) or the threshold itself (build_class(lds_split_min=...))
    if not split or fop not in _SPLIT or nterm <= int(os.environ.get("TC_LDSM_SPLIT_MIN", SPLIT_MIN if split is True else split)):
        return 0
    per = -(-100 // max(aux, 1)) if fop == I.I_sumprod else SPLIT_CHUNK
    return per if nterm > per else 0


def _staged_size(fop, aux, nterm, split=True):
    """staging slots a staged instruction needs (rows padded to a multiple of 8; two-level folds: chunk rows + the
    row of chunk results)"""
    up8 = lambda n_: -(-n_ // 8) * 8         # noqa: E731
    ch = _split_chunk(fop, aux, nterm, split)
    if ch:
        lens = [min(ch, nterm - k) for k in range(0, nterm, ch)]
        return sum(up8(x) for x in lens) + up8(len(lens) + (1 if _SPLIT[fop][1] else 0))
    return up8(nterm) if fop in _PADDED_FOPS else nterm


def build_stream(T, addr, nLds, W=1, stage_base=None, nStage=0, hoist=True, pads=None, one=None, nested=True, split=True):
    """-> dict(words, headers, granges={group: [(first header, end header)]*W}, gstats)

    A LONG instruction (more than 8 operands: the dot products of the factorisation and of the triangular solves,
    norms, minima over all constraints) would keep one lane busy for its whole operand list while the tape words
    of that list trickle in.  With a staging area it is split instead: its TERMS (products, squares, quotients)
    become short instructions of their own -- executed in parallel, one per lane, results in consecutive staging
    slots -- and a FOLD record lets one lane add (min, max) them strictly left to right from those slots, with
    no tape words to wait for.  Every product is rounded exactly as inside the dot (no FMA), and the fold is the
    reference's left-to-right chain, so values do not change."""
    opl, ost = T.ops.tolist(), T.opStart.tolist()
    types = T.types
    nested = bool(nested and hoist and one is not None)
    lvl, nlev = stream_levels(T, nested)
    words, headers = [], []
    granges, gstats = {}, {}
    two_step = {}

    def up8(n_):
        return -(-n_ // 8) * 8

    def chunks_of(s_):
        """[(first term, length)] of the chunks of a two-level fold"""
        n_, ch = len(s_["terms"]), s_["split"]
        return [(k, min(ch, n_ - k)) for k in range(0, n_, ch)]

    def rowlen(s_):
        n_ = len(s_["terms"])
        if s_.get("split"):
            cks = chunks_of(s_)
            return sum(up8(ln) for _, ln in cks) + up8(len(cks) + (1 if _SPLIT[s_["fop"]][1] else 0))
        # + SKEW: rows padded to multiples of 8 doubles all start on the same two LDS banks (64-byte granules of a
        # 128-byte bank period), and the lanes of a fold record read THEIR rows with the same instruction: ten rows, a
        # five-way conflict on every read of the chain.  Two doubles between rows spread the starts over all banks
        # (measured on mpc_quadcopter: a fold step cost 36 cycles per term -- 8 of them the addition).
        return (up8(n_) if s_["fop"] in _PADDED_FOPS else n_) + SKEW
    SKEW = int(os.environ.get("TC_LDSM_ROW_SKEW", "2"))
    LATE_LAST_ON = os.environ.get("TC_LDSM_LATE_LAST", "1") != "0"
    PADS_AS_PRODUCTS = os.environ.get("TC_LDSM_PAD_PRODUCTS", "1") != "0"
    MERGE_PRODUCTS = os.environ.get("TC_LDSM_MERGE_PRODUCTS", "1") != "0"
    MERGE_FOLDS = os.environ.get("TC_LDSM_MERGE_FOLDS", "0") == "1"
    # staging area = two alternating halves for terms computed one step ahead + the classic region, which must hold
    # the longest staged instruction of the class
    longest = 0
    for i in range(1, T.n + 1):
        if T.used[i] and ost[i] - ost[i - 1] > NSHORT:
            op = int(types[i - 1])
            tn = int(T.pars[T.parStart[i - 1]]) if op == I.I_sumprod else (_TERM[op][1] if op in _TERM else 1)
            nt = (ost[i] - ost[i - 1] - _DOT_SKIP.get(op, 0)) // max(tn, 1)
            fop = I.I_sumprod if op == I.I_sumprod else (_TERM[op][2] if op in _TERM else None)
            sz = _staged_size(fop, tn, nt, split) + SKEW
            if sz <= nStage:
                longest = max(longest, sz)
    # TC_LDSM_PREFIX=k (0 = off): a hoisted dot whose first >= k terms were complete THREE levels up has those terms staged two
    # steps ahead and folded one step ahead (an ordinary F_SUM fold, by another lane, usually on another wavefront) into the
    # FIRST slot of its row: the fold on the level's critical path then adds up 1 + the remaining terms -- (-0.0) + partial sum
    # is the partial sum, so the chain is the reference's.  The area behind the classic region is then four buffers (rows and
    # prefix rows, two alternating each) instead of two.
    PREFIX_MIN = int(os.environ.get("TC_LDSM_PREFIX", "0")) if hoist else 0
    if PREFIX_MIN:
        half = (max(0, nStage - longest) // 4) // 8 * 8
        phalf = half
    else:
        half = (max(0, nStage - longest) // 2) // 8 * 8 if hoist else 0
        phalf = 0
    pbase = (stage_base or 0) + 2 * half
    cbase, cstage = pbase + 2 * phalf, nStage - 2 * half - 2 * phalf
    for g, members in enumerate(T.groups, 1):
        levels = {}
        for i in members.tolist():
            if T.used[i]:
                levels.setdefault(lvl[i], []).append(i)
        streams = [[] for _ in range(W)]

        def emit(recs):
            """one step: records spread over the waves (longest first onto the least loaded), then a barrier"""
            if not recs:
                return
            load = [0] * W
            per = [[] for _ in range(W)]
            for r in sorted(recs, key=lambda r: -r[0]):
                k = load.index(min(load))
                load[k] += r[0]
                per[k].append(r)
            # the barrier that ends the step must make HBM writes visible too only if the step has any
            gdst = any(d >= nLds for r in recs for d, _ in r[5])
            for k in range(W):
                first = True
                for q, (_, op, kind, nn, aux, part) in enumerate(per[k]):
                    bar = 0
                    if W > 1 and q == len(per[k]) - 1:       # end of the step: the barrier rides on the wave's last record
                        bar = F_BARRIER_AFTER | (F_BARRIER_GLOBAL if gdst else 0)
                    streams[k].append((op, kind, nn, aux, part, first, bar))
                    first = False
                if W > 1 and not per[k]:
                    streams[k].append(("barrier", gdst))    # a wave without work in this step only waits

        # what a fold record costs its wavefront next to a short one (nn + 6), in the same unit: the level trace of
        # mpc_quadcopter (scripts/level_trace.py) puts a step of one short record at ~600 cycles and a fold of 24 terms
        # with late terms at ~1 300 -- the wavefront that folds should get nothing else until the others hold three shorts
        FOLD_COST = lambda nn, it: it[1][0] + 30       # noqa: E731

        def merge_products(terms):
            """plain products a*b that do not fill their last record ride in the record of the nested ones, as a*(b*1.0),
            when that saves a record (x * 1.0 is x)"""
            k2, k3 = (I.I_sumprod, 2, 2), (OP_PROD3, 3, 0)
            if MERGE_PRODUCTS and one is not None and k2 in terms and k3 in terms:
                n2, n3 = len(terms[k2]), len(terms[k3])
                r2 = n2 % LANES
                if r2 and -(-n2 // LANES) + -(-n3 // LANES) > -(-(n2 - r2) // LANES) + -(-(n3 + r2) // LANES):
                    moved = terms[k2][n2 - r2:]
                    terms[k2] = terms[k2][:n2 - r2]
                    if not terms[k2]:
                        del terms[k2]
                    terms[k3] = terms[k3] + [(d, list(a) + [one * 2]) for d, a in moved]
            return terms

        def merge_short_dots(short):
            """short dot products of one opcode but different lengths share a record when that saves one: the shorter
            ones are padded with products (-0.0) * 1.0 behind their last term -- x + (-0.0) is x"""
            if not (MERGE_PRODUCTS and one is not None and pads is not None):
                return short
            for op in _DOTS:
                keys = [k_ for k_ in short if k_[0] == op]
                if len(keys) < 2:
                    continue
                total = sum(len(short[k_]) for k_ in keys)
                if -(-total // LANES) >= sum(-(-len(short[k_]) // LANES) for k_ in keys):
                    continue
                nmax_ = max(k_[1] for k_ in keys)
                if any((nmax_ - k_[1]) % 2 for k_ in keys):
                    continue
                items = []
                for k_ in sorted(keys):
                    for d, a in short.pop(k_):
                        items.append((d, list(a) + [pads[0] * 2, one * 2] * ((nmax_ - k_[1]) // 2)))
                short[(op, nmax_, 0)] = items
            return short

        def merge_fold_dots(folds):
            """hoisted folds -(acc) join the record of the folds h0 - (acc) of the same step when that saves a record, as
            (-0.0) - (acc): the same value for every acc that is a number (signed zeros included); for a NaN the sign
            bit of the NaN differs, which no instruction of the ISA reads (the host and the device differ in the sign of
            the NaNs they generate anyway).  Likewise (-(acc)) / d as ((-0.0) - (acc)) / d."""
            if not (MERGE_FOLDS and pads is not None):
                return folds
            for plain, full in ((I.I_minus_dot, I.I_plus_minus_dot), (I.I_minus_dot_div, I.I_plus_minus_dot_div)):
                for key in [k_ for k_ in folds if k_[0] == plain]:
                    kf = (full,) + key[1:]
                    if kf in folds and -(-(len(folds[key]) + len(folds[kf])) // LANES) < -(-len(folds[key]) // LANES) + -(-len(folds[kf]) // LANES):
                        for d, body in folds.pop(key):
                            body = list(body)
                            body[1:3] = [pads[0] * 2, body[1]]
                            folds[kf].append((d, body))
            return folds

        def cut(groups_, kind, cost):
            recs = []
            for (op, nn, aux), items in sorted(groups_.items()):
                for k0 in range(0, len(items), LANES):
                    recs.append((cost(nn, items[k0]), op, kind, nn, aux, items[k0:k0 + LANES]))
            return recs
        # ---- classify every level; decide which levels run in ONE step ------------------------------------------
        # Classic level with staged instructions: step A = its short records + ALL terms, step B = the folds (more
        # rounds if the terms do not fit the staging area).  Hoisted level: the terms whose operands were complete
        # two levels up ("early": 83 % of the products of mpc_quadcopter's factorisation) are computed one step
        # AHEAD, next to the folds of the level before, into one of two alternating staging halves; the few "late"
        # terms (an operand from the level just before: at most 2 per instruction, two-factor products) are computed
        # by the folding lane itself, into its own staging slots, right before its chain.  Same operations, same
        # order of additions -- one step (one barrier) per level instead of two.
        Ls = sorted(levels)
        info = []
        for idx, l_ in enumerate(Ls):
            prev = Ls[idx - 1] if idx else None
            short, long_, staged = {}, {}, []
            for i in levels[l_]:
                op = int(types[i - 1])
                ops = opl[ost[i - 1]:ost[i]]
                par = T.pars[T.parStart[i - 1]:T.parStart[i]]
                aux = 0
                if op == I.I_sum:
                    a = [int(addr[j]) * 2 + (0 if p > 0 else 1) for j, p in zip(ops, par)]
                elif op == I.I_sumprod:
                    aux = int(par[0])
                    a = [int(addr[j]) * 2 for j in ops]
                elif op == I.I_componentwise:
                    eq = I.componentwise_opcode(par)
                    if eq is None:
                        raise NotImplementedError("table-driven back-ends run I_componentwise only for plain calls of "
                                                  + str(sorted(I.COMPONENTWISE_CALLS)) + "; got " + repr(I.componentwise_text(par)))
                    op = eq
                    a = [int(addr[j]) * 2 for j in ops]
                elif op > I.I_componentwise or op in (I.I_set, I.I_load):
                    raise NotImplementedError(f"ldsm back-end: opcode {I.NAMES.get(op, op)}")
                else:
                    a = [int(addr[j]) * 2 for j in ops]
                if min(a, default=0) < 0 or addr[i] < 0:
                    raise AssertionError(f"instruction {i}: operand or result without storage")
                d = int(addr[i])
                if not a:
                    a = [d * 2]
                if len(a) <= NSHORT:
                    short.setdefault((op, len(a), aux), []).append((d, a))
                    continue
                if op not in _LONG_OPS:
                    raise NotImplementedError(f"ldsm back-end: {len(a)} operands for opcode {I.NAMES.get(op, op)}")
                # terms of the instruction
                if op == I.I_sumprod:
                    top, tn, fop = "prod", aux, I.I_sumprod
                else:
                    top, tn, fop = _TERM[op]
                skip = _DOT_SKIP.get(op, 0)
                nterm = (len(a) - skip) // tn
                if nStage and tn <= NSHORT and _staged_size(fop, aux, nterm, split) + SKEW <= cstage and (len(a) - skip) % tn == 0:
                    tl, late, oldt = [], [], []
                    for q in range(nterm):
                        ids = list(ops[skip + q * tn:skip + (q + 1) * tn])
                        tw = list(a[skip + q * tn:skip + (q + 1) * tn])
                        if nested and op in _DOTS:
                            # a factor that is a plain product of the level before (or later: stream_levels did not
                            # wait for it) is expanded: a*v -> a*(p*q), the same two roundings in the same order
                            k = _nested_factor(T, opl, ost, g, ids)
                            if k is not None and lvl[ids[k]] >= (prev if prev is not None else l_):
                                dn = _div_nest(T, opl, ost, g, ids, k)
                                pj = opl[ost[ids[k] - 1]:ost[ids[k]]]
                                ids = [ids[1 - k]] + list(pj)
                                tw = [tw[1 - k]] + [int(addr[j]) * 2 for j in pj]
                                if dn is not None and any(T.group[j] == g and T.used[j] and lvl[j] >= l_ for j in ids):
                                    # the quotients themselves are values of THIS level (stream_levels counted the term
                                    # through N1, N2, D): the folding lane divides
                                    ids = list(dn[:3])
                                    tw = _DN(int(addr[j]) * 2 for j in ids)
                                if min(tw) < 0:
                                    raise AssertionError("nested product: operand without storage")
                        tl.append(tw)
                        late.append(prev is not None and any(T.group[j] == g and T.used[j] and lvl[j] >= prev for j in ids))
                        oldt.append(idx >= 2 and not any(T.group[j] == g and T.used[j] and lvl[j] >= Ls[idx - 2] for j in ids))
                        if any(T.group[j] == g and T.used[j] and lvl[j] >= l_ for j in ids):
                            raise AssertionError(f"instruction {i}: a term needs a value of its own level")
                    s_ = dict(op=op, fop=fop, aux=aux, d=d, heads=a[:skip], top=top, tn=tn, terms=tl, late=late, old=oldt)
                    # A long minimum / maximum does not depend on the order of its operands except through NaNs and
                    # the signs of zeros, and a long sum of products is DEFINED by the reference as a sum of
                    # statements of <= MAX_TERMS_PERLINE factors (instructionsTableUTHash.c:1003-1012): the chunks
                    # (the reference's own lines for the sum) are folded by one lane each, a second fold combines
                    # the chunk results in order.  Chunks of a minimum start from +inf (a NaN is skipped as in the
                    # reference's chain); where the reference starts from its first operand that operand leads the
                    # second level, so a NaN there sticks and the first of equal zeros stays.
                    if _split_chunk(fop, aux, nterm, split):
                        s_["split"] = _split_chunk(fop, aux, nterm, split)
                    staged.append(s_)
                else:
                    long_.setdefault((op, 0, aux), []).append((d, a))
            for key, items in long_.items():
                long_[key] = sorted(items, key=lambda it: -len(it[1]))      # similar lengths share a record
            staged.sort(key=lambda s_: -len(s_["terms"]))
            hoist = bool(staged) and idx > 0 and half >= 64 and not any(s_.get("split") for s_ in staged)
            why = None if hoist or not staged else ("first level" if idx == 0 else "split row" if any(s_.get("split") for s_ in staged)
                                                    else "no staging halves")
            for s_ in staged:
                nl = sum(s_["late"])
                if nl and not (s_["top"] == "prod" and s_["tn"] == 2 and nl <= NLATE):
                    hoist, why = False, why or ("more than %d late terms" % NLATE if nl > NLATE else "late term that is not a product")
            if hoist and PREFIX_MIN and idx >= 2:
                # prefixes, longest first, while the prefix rows fit their buffer
                cand = []
                for s_ in staged:
                    if s_["top"] == "prod" and s_["tn"] == 2 and s_["fop"] in _DOT_SKIP and not s_.get("split"):
                        pf = 0
                        while pf < len(s_["terms"]) and s_["old"][pf] and not s_["late"][pf]:
                            pf += 1
                        if pf >= PREFIX_MIN:
                            cand.append((pf, s_))
                used_p = 0
                for pf, s_ in sorted(cand, key=lambda c_: -c_[0]):
                    if used_p + up8(pf) + SKEW <= phalf:
                        used_p += up8(pf) + SKEW
                        s_["pfx_terms"] = s_["terms"][:pf]
                        s_["terms"] = [None] + s_["terms"][pf:]          # None: the slot the prefix fold writes
                        s_["late"] = [False] + s_["late"][pf:]
            if sum(rowlen(s_) for s_ in staged) > half:            # (late terms keep their slot in the row)
                hoist, why = False, why or "rows exceed a staging half"
            if not hoist:
                for s_ in staged:                                   # (a level that takes two steps after all: rows as they were)
                    if "pfx_terms" in s_:
                        s_["terms"] = s_.pop("pfx_terms") + s_["terms"][1:]
                        s_["late"] = [False] * (len(s_["terms"]) - len(s_["late"]) + 1) + s_["late"][1:]
            if why:                       # diagnostics: why a level with staged rows takes two steps (class_stream()["two_step_levels"])
                two_step[why] = two_step.get(why, 0) + 1
            info.append(dict(short=short, long_=long_, staged=staged, hoist=hoist))

        def late_last(s_):
            """the row's late terms are its last terms (or it has none) and it is an ordered sum of products"""
            lt = [q for q, x in enumerate(s_["late"]) if x]
            n_ = len(s_["terms"])
            return (LATE_LAST_ON and s_["fop"] in _DOT_SKIP and not s_.get("split") and n_ - len(lt) >= 1
                    and lt == list(range(n_ - len(lt), n_)))

        def fold_item(s_, base, inline):
            hw = [s_["heads"][q] if q < len(s_["heads"]) else 2 for q in range(2)]
            body = [len(s_["terms"]), hw[0], hw[1], base]
            if inline:
                lt = [q for q, x in enumerate(s_["late"]) if x]
                ll = late_last(s_)
                if ll:
                    body[0] -= len(lt)      # the chain runs over the early terms; the late products follow it
                body.append(len(lt))
                for q in range(NLATE):      # position, a, b, c: the term is a * (b * c)
                    tw = s_["terms"][lt[q]] if q < len(lt) else [2, 2]
                    body += [lt[q] if q < len(lt) and not ll else 0, tw[0], tw[1], tw[2] if len(tw) == 3 else one * 2]
            return (s_["d"], body)

        def term_key(s_, tw):
            if s_["top"] == "prod":
                if isinstance(tw, _DN):
                    return (OP_PRODDN, 3, 0)
                return (OP_PROD3, 3, 0) if len(tw) == 3 and s_["tn"] == 2 else (I.I_sumprod, s_["tn"], s_["tn"])
            return (s_["top"], s_["tn"], 0)

        def dn_bits(s_):
            """which of the (at most NLATE) inline late terms of a hoisted row are of the OP_PRODDN form"""
            lt = [q for q, x in enumerate(s_["late"]) if x]
            return sum(1 << q for q in range(min(len(lt), NLATE)) if isinstance(s_["terms"][lt[q]], _DN))

        def pad_row(terms, fop, base, n_, like=None):
            """the operation's identity (-0.0 for sums, +inf for minima, -inf for maxima) into the slots between the
            end of a row of n_ terms and the next multiple of 8.  like = the key of the row's own two-factor products:
            the pad is then written as identity * 1.0 by the record that computes those products (one record less in
            nearly every step of a factorisation; -0.0 * 1.0 is -0.0, inf * 1.0 is inf)"""
            if fop in _PADDED_FOPS:
                for q in range(n_, up8(n_)):
                    pw = pads[_PADDED_FOPS[fop]] * 2
                    if like is not None and PADS_AS_PRODUCTS:
                        terms.setdefault(like, []).append((base + q, [pw, one * 2]))
                    else:
                        terms.setdefault((OP_COPY, 1, 0), []).append((base + q, [pw]))

        def add_pads(terms, s_, base, hoisted=False):
            n_ = len(s_["terms"])
            if hoisted and late_last(s_):
                n_ -= sum(s_["late"])       # the late slots are never written: the identity from the first of them on
            like = None
            if one is not None and s_["top"] == "prod" and s_["tn"] == 2 and not s_.get("split"):
                plain = [tw for q, tw in enumerate(s_["terms"]) if tw is not None and len(tw) == 2 and not (hoisted and s_["late"][q])]
                if plain:                    # the row has plain products staged in this very step
                    like = term_key(s_, plain[0])
            pad_row(terms, s_["fop"], base, n_, like)
        toggle = 0               # which staging half the next hoisted level's early terms go to
        ptoggle = 0              # ... and which prefix buffer the level after that one's prefix rows go to
        for idx, lv in enumerate(info):
            recs = cut(merge_short_dots(lv["short"]), K_SHORT, lambda nn, it: nn + 6) + cut(lv["long_"], K_LONG, lambda nn, it: len(it[1]) + 8)
            nxt = info[idx + 1] if idx + 1 < len(info) else None
            ahead_next = []
            if nxt is not None and nxt["hoist"]:
                terms, used = {}, 0
                hb = stage_base + toggle * half
                toggle ^= 1
                pfolds = {}
                for s_ in nxt["staged"]:
                    s_["base"] = hb + used
                    for q, tw in enumerate(s_["terms"]):
                        if tw is not None and not s_["late"][q]:
                            terms.setdefault(term_key(s_, tw), []).append((s_["base"] + q, tw))
                    if "pfx_terms" in s_:          # its prefix row was staged a step ago: folded now, into the row's first slot
                        pfolds.setdefault((F_SUM, 0, 0), []).append((s_["base"], [len(s_["pfx_terms"]), 2, 2, s_["pbase"]]))
                    add_pads(terms, s_, s_["base"], True)
                    used += rowlen(s_)             # the late terms keep their slot in the row: the folding lane fills it
                if used > half:
                    raise AssertionError("staging half overflow")
                for key, items in pfolds.items():
                    pfolds[key] = sorted(items, key=lambda it: -it[1][0])
                ahead_next = cut(merge_products(terms), K_SHORT, lambda nn, it: nn + 6) + cut(pfolds, K_FOLD, FOLD_COST)
            nxt2 = info[idx + 2] if idx + 2 < len(info) else None
            if nxt2 is not None and nxt2["hoist"] and any("pfx_terms" in s_ for s_ in nxt2["staged"]):
                terms, used = {}, 0
                pb = pbase + ptoggle * phalf
                ptoggle ^= 1
                for s_ in nxt2["staged"]:
                    if "pfx_terms" in s_:
                        s_["pbase"] = pb + used
                        for q, tw in enumerate(s_["pfx_terms"]):
                            terms.setdefault(term_key(s_, tw), []).append((s_["pbase"] + q, tw))
                        pad_row(terms, F_SUM, s_["pbase"], len(s_["pfx_terms"]), term_key(s_, s_["pfx_terms"][0]))
                        used += up8(len(s_["pfx_terms"])) + SKEW
                if used > phalf:
                    raise AssertionError("prefix buffer overflow")
                ahead_next = ahead_next + cut(merge_products(terms), K_SHORT, lambda nn, it: nn + 6)
            staged = lv["staged"]
            if lv["hoist"]:
                folds = {}
                for s_ in staged:
                    folds.setdefault((s_["fop"], 0, s_["aux"] | (LATE_LAST if late_last(s_) else 0) | (dn_bits(s_) << DN_SHIFT)),
                                     []).append(fold_item(s_, s_["base"], True))
                folds = merge_fold_dots(folds)
                for key, items in folds.items():
                    folds[key] = sorted(items, key=lambda it: -it[1][0])
                emit(recs + cut(folds, K_FOLD, FOLD_COST) + ahead_next)
            elif not staged:
                emit(recs + ahead_next)
            else:
                # rounds that fit the classic staging region; the first round's terms share the step of the level's
                # short records, the next level's early terms the step of the last folds
                k0 = 0
                while True:
                    terms, folds, used = {}, {}, 0
                    extra, finals = {}, {}      # two-level folds: copies / pads beside the chunk folds, second level
                    while k0 < len(staged) and used + rowlen(staged[k0]) <= cstage:
                        s_ = staged[k0]
                        base = cbase + used
                        if s_.get("split"):
                            fop1, lead = _SPLIT[s_["fop"]]
                            cks = chunks_of(s_)
                            row2 = base + sum(up8(ln) for _, ln in cks)
                            off2 = 1 if lead else 0
                            b_ = base
                            for k, (t0, ln) in enumerate(cks):
                                for q in range(ln):
                                    tw = s_["terms"][t0 + q]
                                    terms.setdefault(term_key(s_, tw), []).append((b_ + q, tw))
                                pad_row(terms, fop1, b_, ln)
                                folds.setdefault((fop1, 0, 0), []).append((row2 + off2 + k, [ln, 2, 2, b_]))
                                b_ += up8(ln)
                            if lead:
                                extra.setdefault((OP_COPY, 1, 0), []).append((row2, [base * 2]))
                            pad_row(extra, s_["fop"], row2, len(cks) + off2)
                            fop2 = F_SUM if s_["fop"] == I.I_sumprod else s_["fop"]
                            finals.setdefault((fop2, 0, 0), []).append((s_["d"], [len(cks) + off2, 2, 2, row2]))
                        else:
                            for q, tw in enumerate(s_["terms"]):
                                terms.setdefault(term_key(s_, tw), []).append((base + q, tw))
                            add_pads(terms, s_, base)
                            folds.setdefault((s_["fop"], 0, s_["aux"]), []).append(fold_item(s_, base, False))
                        used += rowlen(s_)
                        k0 += 1
                    emit(recs + cut(merge_products(terms), K_SHORT, lambda nn, it: nn + 6))
                    recs = []
                    for key, items in folds.items():
                        folds[key] = sorted(items, key=lambda it: -it[1][0])
                    last = k0 >= len(staged)
                    emit(cut(folds, K_FOLD, FOLD_COST) + cut(extra, K_SHORT, lambda nn, it: nn + 6)
                         + (ahead_next if last and not finals else []))
                    if finals:
                        emit(cut(finals, K_FOLD, FOLD_COST) + (ahead_next if last else []))
                    if last:
                        break
        rg = []
        for k in range(W):
            h0 = len(headers) // 8
            for r in streams[k]:
                if r[0] == "barrier":
                    headers += [K_BARRIER << 8, 0, 0, 0, 0, F_NEW_LEVEL | (F_GLOBAL_DST if r[1] else 0), 0, 0]
                else:
                    op, kind, nn, aux, part, first, bar = r
                    _rec(words, headers, op, kind, nn, aux, part, nLds, first)
                    headers[-3] |= bar
            rg.append((h0, len(headers) // 8))
        granges[g] = rg
        gstats[g] = (len(members), len(levels))
    return dict(words=words, headers=headers, granges=granges, gstats=gstats, two_step_levels=two_step)


def copy_records(words, headers, pairs, nLds, W=1):
    """dest <- source for a copy entry point (pairs of addresses) -> [(first header, end header)]*W"""
    items = [(int(d), [int(s) * 2]) for d, s in pairs]
    rg = []
    if set(d for d, _ in items) & set(a[0] >> 1 for _, a in items):
        chunks = [[it] for it in items]           # keep the sequential element-by-element meaning
        chunks = [chunks] + [[] for _ in range(W - 1)]
        seq = True
    else:
        recs = [items[k0:k0 + LANES] for k0 in range(0, len(items), LANES)]
        chunks = [recs[k::W] for k in range(W)]
        seq = False
    for k in range(W):
        h0 = len(headers) // 8
        for part in chunks[k]:
            _rec(words, headers, OP_COPY, K_SHORT, 1, 0, part, nLds, seq)
        if W > 1:
            headers += [K_BARRIER << 8, 0, 0, 0, 0, F_NEW_LEVEL | F_GLOBAL_DST, 0, 0]
        rg.append((h0, len(headers) // 8))
    return rg


def class_stream(T, classname, defines, W=1, lds_slots=20000, hoist=True, nested=True, split=True, stage=0):
    """everything the device walk needs for one class"""
    # the life times behind the packing of group-local values are those of the DEFAULT schedule (terms staged one step ahead,
    # nested products): the diagnostic schedules (hoist=False, nested=False: tests/test_ldsm_stream.py) read later than that
    # and give every local value a slot of its own
    sl = assign_slots(T, defines, lds_slots, stage=int(stage) or STAGE, pack_locals=None if (hoist and nested) else False)
    addr, nLds = sl["addr"], sl["nLds"]
    st = build_stream(T, addr, nLds, W, stage_base=sl["stage_base"], nStage=sl["nStage"], hoist=hoist,
                      pads=(sl["negzero"], sl["posinf"], sl["neginf"]), one=sl["one"], nested=nested, split=split)
    words, headers = st["words"], st["headers"]
    rid, table = {}, []
    for g in sorted(st["granges"]):
        rid[g] = len(table) // (2 * W)
        table += [x for se in st["granges"][g] for x in se]
    for e in T.copies:
        pairs = [(addr[d], addr[s_]) for d, s_ in zip(e["dst"], e["src"]) if addr[d] >= 0]
        if any(a < 0 for _, a in pairs):
            raise AssertionError("copy source without storage")
        rg = copy_records(words, headers, pairs, nLds, W)
        rid[e["name"]] = len(table) // (2 * W)
        table += [x for se in rg for x in se]
    go = T.entry(f"{classname}_getOutputs")[2]
    out_addr = [int(addr[i]) for o in go["outputs"] for i in o["ins"]]
    if out_addr and min(out_addr) < 0:
        raise AssertionError("output without storage")
    out_off = len(words)
    words += out_addr + [0] * (-len(out_addr) % 8)
    unwalked = {rid[g] for g in getattr(sl["lowering"], "dead_groups", ()) if g in rid}
    validate(words, headers, table, W, nLds, nLds + sl["nGlob"], dummy=sl["dummy"], unwalked=unwalked)
    return dict(sl, words=words, headers=headers, rid=rid, table=table, out_off=out_off, nOut=len(out_addr),
                gstats=st["gstats"], W=W, two_step_levels=st["two_step_levels"])


def record_accesses(words, hdr):
    """(addresses read, addresses written) by the active lanes of one record -- the device walk's semantics, as the
    CPU model of tests/test_ldsm_stream.py restates them"""
    w0, n, aux, cnt, off, flags, cn, _ = (int(x) for x in hdr)
    kind, op = w0 >> 8, w0 & 255
    body = record_body(words, hdr)
    rd, wr = set(), set(int(x) for x in body[0][:cnt])
    if kind == K_SHORT:
        for j in range(n):
            rd.update(int(x) >> 1 for x in body[1 + j][:cnt])
    elif kind == K_FOLD:
        nlm = (flags >> F_NLATE_SHIFT) & 3
        for q in range(cnt):
            ln, base = int(body[1][q]), int(body[4][q])
            nrow = -(-ln // 8) * 8 if flags & F_PADDED else ln
            rd.update(range(base, base + nrow))
            rd.add(int(body[2][q]) >> 1)
            rd.add(int(body[3][q]) >> 1)
            if nlm:
                for t in range(int(body[5][q])):
                    if not (aux & LATE_LAST and op in _DOT_SKIP):
                        wr.add(base + int(body[6 + 4 * t][q]))
                    rd.update(int(body[7 + 4 * t + j][q]) >> 1 for j in range(3))
    else:
        skip = _DOT_SKIP.get(op, 0)
        for q in range(cnt):
            rd.update(int(body[2 + j][q]) >> 1 for j in range(skip))
            rd.update(int(body[4 + k][q]) >> 1 for k in range(int(body[1][q])))
    return rd, wr


def validate(words, headers, table, W, nLds, nAddr, dummy=None, unwalked=()):
    """every address a lane would touch is inside [0, nAddr); ranges end on header boundaries; the waves sharing a
    range meet the same number of barriers; and within one step (barrier to barrier) no wave writes an address that
    ANOTHER wave of the range reads or writes in the same step -- on the device the waves of a step run side by
    side, so such a pair would be a race whatever order a sequential model happens to take (`dummy`: the slot the
    masked lanes of ragged records write to, never read for its value; `unwalked`: ranges of groups that no entry
    point of the class ever evaluates -- their storage was packed without regard to them, Lowering.dead_groups)"""
    nh = len(headers) // 8
    for r0 in range(0, len(table), 2 * W):
        if W > 1 and r0 // (2 * W) not in unwalked:
            steps = []                                    # per wave: [(reads, writes) per step]
            for k in range(W):
                cur, mine = (set(), set()), []
                for h in range(table[r0 + 2 * k], table[r0 + 2 * k + 1]):
                    hdr = headers[8 * h:8 * h + 8]
                    if (hdr[0] >> 8) == K_BARRIER:
                        mine.append(cur)
                        cur = (set(), set())
                        continue
                    rd, wr = record_accesses(words, hdr)
                    cur[0].update(rd)
                    cur[1].update(wr)
                    if hdr[5] & F_BARRIER_AFTER:
                        mine.append(cur)
                        cur = (set(), set())
                mine.append(cur)
                steps.append(mine)
            for st in range(max(len(m_) for m_ in steps)):
                acc = [m_[st] if st < len(m_) else (set(), set()) for m_ in steps]
                for k in range(W):
                    wk = acc[k][1] - {dummy}
                    if not wk:
                        continue
                    for j in range(W):
                        if j != k and (wk & acc[j][0] or (j > k and wk & (acc[j][1] - {dummy}))):
                            raise AssertionError("step %d of range %d: wave %d writes what wave %d touches (%s)"
                                                 % (st, r0 // (2 * W), k, j, sorted(wk & (acc[j][0] | acc[j][1]))[:4]))
        bars = set()
        for k in range(W):
            h0, h1 = table[r0 + 2 * k], table[r0 + 2 * k + 1]
            if not (0 <= h0 <= h1 <= nh):
                raise AssertionError("bad header range")
            nb = 0
            for h in range(h0, h1):
                w0, n, aux, cnt, off, flags, cn, _ = headers[8 * h:8 * h + 8]
                kind = w0 >> 8
                if kind == K_BARRIER:
                    nb += 1
                    continue
                if flags & F_BARRIER_AFTER:
                    nb += 1
                if not (1 <= cnt <= LANES and cn % 8 == 0 and cnt <= cn):
                    raise AssertionError("bad record count")
                nlm = (flags >> F_NLATE_SHIFT) & 3 if kind == K_FOLD else 0
                body = record_body(words, headers[8 * h:8 * h + 8])
                dst = body[0]
                if kind == K_FOLD:
                    rl = (body[1] + 7) // 8 * 8 if flags & F_PADDED else body[1]
                    if n % 8 or body[1].max() > n or body[1][:cnt].min() < 1 or body[4].min() < 0 or \
                            (body[4] + rl).max() > nLds:
                        raise AssertionError("bad fold record")
                    ops = body[2:4] >> 1
                    if nlm:
                        if nlm > NLATE or body[5].max() != nlm or body[5].min() < 0:
                            raise AssertionError("bad late-term count")
                        for q in range(NLATE):
                            act = body[5][:cnt] > q
                            if (body[6 + 4 * q][:cnt][act] >= body[1][:cnt][act]).any() or body[6 + 4 * q].min() < 0:
                                raise AssertionError("late term outside the row")
                        ops = np.concatenate([ops] + [body[7 + 4 * q:10 + 4 * q] >> 1 for q in range(NLATE)])
                else:
                    ops = body[1:1 + n] >> 1 if kind == K_SHORT else body[2:4 + n] >> 1
                if dst.min() < 0 or dst.max() >= nAddr or ops.min() < 0 or ops.max() >= nAddr:
                    raise AssertionError("address outside the state")
                if kind == K_SHORT and not (1 <= n <= NSHORT):
                    raise AssertionError("bad short record")
                if kind == K_LONG and (n % 8 or body[1].max() > n or body[1][:cnt].min() < 1):
                    raise AssertionError("bad long record")
                gsrc = bool((ops >= nLds).any())
                gdst = bool((dst >= nLds).any())
                if gsrc != bool(flags & F_GLOBAL_SRC) or gdst != bool(flags & F_GLOBAL_DST):
                    raise AssertionError("global flags do not match the addresses")
            bars.add(nb)
        if len(bars) > 1:
            raise AssertionError("waves of one range do not meet the same number of barriers")


def as_array(words):
    a = np.asarray(words, dtype=np.int64)
    if a.size and (a.max() >= 2 ** 31 or a.min() < -2 ** 31):
        raise OverflowError("stream word out of int32 range")
    return a.astype(np.int32)
