"""csparse: vectorized tensor operations -> deduplicated scalar instruction tape.

Host-side (code-generation time) restatement of the reference's ``@csparse`` class:

* ``declareSet/Get/Copy/Alias``            lib/@csparse/csparse.m:296-650
* per-operator structural scalarizers      lib/@csparse/sparsity_*.m,
                                           computeScalarInstructions.m:165-433
* operand canonicalisation before CSE      csparse.m:1168-1227 (``normalize``)
* symbolic LDL / triangular solves         sparsity_ldl.m:161-222,
                                           sparsity_mldivide_{l1,d,u1}.m
* dependency groups                        dependencyGroups.m:36-455

Design differences (deliberate, MI355X-first): sparse tensors are numpy index/instruction
arrays joined with vectorised hash-joins instead of MATLAB ``ismember``/``kron`` expansion;
the fill-reducing LDL order is our own minimum-degree ordering (MATLAB's ``symamd`` is not
available); dependency signatures are Python-int bitmasks; and the result is a
:class:`Tape` object that the HIP emitter, the CPU oracle and the computation-graph exporter
all consume.
"""
from __future__ import annotations

import os

import numpy as np

from . import instructions as I
from . import tcalculus as tc
from .instructions import InstructionsTable

__all__ = ["CSparse", "Tape", "min_degree_order", "multiple_elimination_order", "depth_aware_order", "symbolic_ldl_cost"]


# ----------------------------------------------------------------------------------
# sparse-tensor helpers: (subs[nnz, ndim] int64 0-based, ins[nnz] int64 1-based ids)
# ----------------------------------------------------------------------------------
def _empty(ndim):
    return np.zeros((0, ndim), dtype=np.int64), np.zeros(0, dtype=np.int64)


def _lin(subs, size):
    """column-major linear index"""
    if subs.shape[1] == 0:
        return np.zeros(subs.shape[0], dtype=np.int64)
    return np.ravel_multi_index(tuple(subs.T), tuple(size), order="F").astype(np.int64)


def _unlin(lin, size):
    if len(size) == 0:
        return np.zeros((lin.size, 0), dtype=np.int64)
    return np.stack(np.unravel_index(lin, tuple(size), order="F"), axis=1).astype(np.int64)


def _sort_cm(subs, ins, size):
    k = np.argsort(_lin(subs, size), kind="stable")
    return subs[k], ins[k]


def _full_subs(size):
    return _unlin(np.arange(tc._prod(size), dtype=np.int64), size)


def _join(keyT, keyS):
    """All (rowT,rowS) pairs with keyT[rowT]==keyS[rowS]."""
    order = np.argsort(keyS, kind="stable")
    ks = keyS[order]
    lo = np.searchsorted(ks, keyT, "left")
    hi = np.searchsorted(ks, keyT, "right")
    cnt = hi - lo
    total = int(cnt.sum())
    rowsT = np.repeat(np.arange(keyT.size, dtype=np.int64), cnt)
    start = np.repeat(lo, cnt)
    within = np.arange(total, dtype=np.int64) - np.repeat(np.cumsum(cnt) - cnt, cnt)
    rowsS = order[start + within]
    return rowsT, rowsS


def _groups(sorted_keys):
    """start offsets and counts of runs of equal keys"""
    n = sorted_keys.size
    if n == 0:
        return np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64)
    brk = np.flatnonzero(np.r_[True, sorted_keys[1:] != sorted_keys[:-1]])
    cnt = np.diff(np.r_[brk, n])
    return brk.astype(np.int64), cnt.astype(np.int64)


LDL_GATE = "one"     # how `stages` constrains the elimination order (see min_degree_order)
# which elimination order `ldl` nodes get: "mindeg" = greedy minimum degree (fill first; stands in for symamd), or
# ("depth", width) = depth_aware_order: the shortest critical path that `width` parallel lanes can pay for in fill.
# CSparse.ldl_order overrides it per generated class (cmex2optimizeCS(..., ldlOrder=...)).
LDL_ORDER = "mindeg"


def min_degree_order(n, rows, cols, stages=None, gate=None):
    """Greedy minimum-degree elimination order of the symmetric pattern (rows,cols).
    Stands in for MATLAB ``symamd(A,[1000,0])`` at sparsity_ldl.m:91; any valid order
    gives the same solution up to rounding (SURVEY.md 7 hard part 1).  Ties are broken
    by lowest index so the order is deterministic.

    ``stages`` marks late nodes (stage>0: the equality duals of a KKT matrix, whose own diagonal
    is only -addEye2HessianEq); ``gate`` says what a late node waits for:
      "stage" every node of stage s is eliminated before any node of stage s+1 (numerically the
              most conservative, but the Schur complement on the duals fills in: 2.1 M
              multiply-adds per factorisation for mpc_quadcopter);
      "all"   a late node becomes eligible once all its original stage-0 neighbours are gone;
      "one"   ... once at least one of them is gone, i.e. as soon as its pivot has received a
              Schur-complement contribution and is no longer the bare -1e-9 (55 k multiply-adds
              for mpc_quadcopter, fill like the unconstrained order).
    Default: module attribute LDL_GATE."""
    gate = gate or LDL_GATE
    adj = [set() for _ in range(n)]
    for r, c in zip(rows, cols):
        r, c = int(r), int(c)
        if r != c:
            adj[r].add(c)
            adj[c].add(r)
    import heapq
    stg = [0] * n if stages is None else [int(s) if int(s) != 2 else 0 for s in stages]    # 2: see multiple_elimination_order
    if gate == "stage":
        key = stg
        need = [0] * n
        gated_by = [()] * n
    elif gate in ("all", "one"):
        key = [0] * n
        need = [0] * n
        gated_by = [[] for _ in range(n)]
        for i in range(n):
            if stg[i]:
                early = [a for a in adj[i] if stg[a] == 0]
                need[i] = len(early) if gate == "all" else min(1, len(early))
                for a in early:
                    gated_by[a].append(i)
    else:
        raise ValueError("min_degree_order: gate must be 'stage', 'all' or 'one'")
    seen = [0] * n                      # eliminated stage-0 neighbours of a late node
    heap = [(key[i], len(adj[i]), i) for i in range(n) if need[i] == 0]
    heapq.heapify(heap)
    done = [False] * n
    order = []
    while heap:
        _, d, i = heapq.heappop(heap)
        if done[i] or d != len(adj[i]):
            continue
        done[i] = True
        order.append(i)
        nb = adj[i]
        for a in nb:
            adj[a].discard(i)
        nbl = list(nb)
        for a in nbl:
            new = nb - adj[a]
            new.discard(a)
            if new:
                adj[a] |= new
            if seen[a] >= need[a]:
                heapq.heappush(heap, (key[a], len(adj[a]), a))
        for a in gated_by[i]:
            seen[a] += 1
            if not done[a] and seen[a] == need[a]:
                heapq.heappush(heap, (key[a], len(adj[a]), a))
        adj[i] = set()
    if len(order) != n:
        raise AssertionError("min_degree_order: gated nodes left over")
    return np.array(order, dtype=np.int64)


def _gate_one(n, adj, stages, eq_gate="one"):
    """(need, gated_by): how many eliminations a node waits for, and who is told when a node goes.  Stage 1 (equality
    duals, own diagonal -addEye2HessianEq) waits for eliminated stage-0 neighbours: ONE of them (`eq_gate` "one", the
    gate of min_degree_order), two ("two"), or all but one ("allbut1") -- the more of them have gone, the more of the
    Schur complement -G H^-1 G' the pivot has received.  Stage 2 marks the inequality duals (own diagonal
    -F*lambda): a stage-0 node waits for ALL its stage-2 neighbours, so that a primal variable whose only curvature is
    the barrier term lambda/F (a slack, a norm's flat direction) never becomes a pivot of size addEye2HessianU before
    it has received that term."""
    stg = [0] * n if stages is None else [int(s) for s in stages]
    need = [0] * n
    gated_by = [[] for _ in range(n)]
    for i in range(n):
        if stg[i] == 1:
            early = [a for a in adj[i] if stg[a] == 0]
            k = len(early)
            need[i] = {"one": min(1, k), "two": min(2, k), "allbut1": max(min(1, k), k - 1)}[eq_gate]
        elif stg[i] == 0:
            early = [a for a in adj[i] if stg[a] == 2]
            need[i] = len(early)
        else:
            early = []
        for a in early:
            gated_by[a].append(i)
    return need, gated_by


def multiple_elimination_order(n, rows, cols, stages=None, slack=0, eq_gate="one"):
    """Elimination order built in ROUNDS: every round eliminates a maximal independent set of the eligible nodes whose
    degree is within ``slack`` of the current minimum (Liu's multiple minimum degree, with a tolerance).  Nodes of one
    round do not depend on each other, so the elimination tree gets short and bushy: a tridiagonal block is reduced
    odd-even (cyclic reduction, depth log n) instead of front to back (depth n).  Fill grows with ``slack``.
    Late nodes (``stages``) obey the gate "one" of :func:`min_degree_order`."""
    adj = [set() for _ in range(n)]
    for r, c in zip(rows, cols):
        r, c = int(r), int(c)
        if r != c:
            adj[r].add(c)
            adj[c].add(r)
    need, gated_by = _gate_one(n, adj, stages, eq_gate)
    seen = [0] * n
    done = [False] * n
    order = []
    while len(order) < n:
        elig = [i for i in range(n) if not done[i] and seen[i] >= need[i]]
        if not elig:
            raise AssertionError("multiple_elimination_order: gated nodes left over")
        dmin = min(len(adj[i]) for i in elig)
        blocked = set()
        chosen = []
        for d, i in sorted((len(adj[i]), i) for i in elig if len(adj[i]) <= dmin + slack):
            if i not in blocked:
                chosen.append(i)
                blocked.add(i)
                blocked |= adj[i]
        for i in chosen:
            done[i] = True
            order.append(i)
            nb = adj[i]
            for a in nb:
                adj[a].discard(i)
            for a in list(nb):
                new = nb - adj[a]
                new.discard(a)
                if new:
                    adj[a] |= new
            for a in gated_by[i]:
                seen[a] += 1
            adj[i] = set()
    return np.array(order, dtype=np.int64)


def symbolic_ldl_cost(n, rows, cols, p, term_weight=1.0 / 16):
    """What the up-looking LDL' of sparsity_ldl.m:161-222 and the two triangular solves cost under the order ``p``:
    nnz(L), multiply-adds, and the length of the longest dependency chain through factorisation + forward + backward
    substitution, an instruction with t terms counting 1 + t*term_weight (its ordered sum is a chain of its own)."""
    pinv = np.empty(n, dtype=np.int64)
    pinv[np.asarray(p)] = np.arange(n)
    acol = [set() for _ in range(n)]            # permuted strictly-lower structure of A by column
    for r, c in zip(rows, cols):
        pr, pc = int(pinv[int(r)]), int(pinv[int(c)])
        if pr > pc:
            acol[pc].add(pr)
        elif pc > pr:
            acol[pr].add(pc)
    Lrow = [dict() for _ in range(n)]           # row -> {col: finish time of L(row, col)}
    Lcol = [[] for _ in range(n)]
    tD = [0.0] * n
    flops = 0
    for col in range(n):
        Lr = Lrow[col]
        iss = sorted(Lr)
        t0 = 0.0
        for i in iss:
            t0 = max(t0, Lr[i], tD[i])
        tD[col] = t0 + 1 + len(iss) * term_weight if iss else 0.0
        flops += 2 * len(iss)
        cand = set(acol[col])
        for i in iss:
            cand.update(r for r in Lcol[i] if r > col)
        for row in sorted(cand):
            Lrr = Lrow[row]
            isr = [i for i in iss if i in Lrr]
            t1 = tD[col]
            for i in isr:
                t1 = max(t1, Lrr[i], Lr[i], tD[i])
            Lrr[col] = t1 + 1 + len(isr) * term_weight
            Lcol[col].append(row)
            flops += 2 * len(isr) + 1
    depth = max(max(tD), max((max(d.values()) for d in Lrow if d), default=0.0))
    y = [0.0] * n
    for r in range(n):
        if Lrow[r]:
            y[r] = max(y[k] for k in Lrow[r]) + 1 + len(Lrow[r]) * term_weight
    x = [0.0] * n
    for r in range(n - 1, -1, -1):
        if Lcol[r]:
            x[r] = max(x[k] for k in Lcol[r]) + 1 + len(Lcol[r]) * term_weight
    flops += 4 * sum(len(d) for d in Lrow)
    return dict(nnzL=sum(len(d) for d in Lrow), flops=flops, depth=depth + max(y) + max(x))


def depth_aware_order(n, rows, cols, stages=None, width=64):
    """The elimination order for a machine that is wide but slow per dependent step: among the greedy minimum-degree
    order and multiple-elimination orders of growing tolerance, the one with the smallest
    ``critical path + multiply-adds / width`` (:func:`symbolic_ldl_cost`).  ``width`` = how many scalar
    instructions of one instance the back-end executes side by side.

    With equality duals (stage 1) in the matrix only tolerance 0 is tried: their pivots are the bare
    -addEye2HessianEq until a neighbour has been eliminated, and which neighbour comes first decides how many digits
    survive -- measured on mpc_quadcopter's and mpc_dcmotor's KKT matrices, tolerance >= 1 (quadcopter) or >= 2
    (dcmotor) turns a direction error of 1e-9 / 1e-16 into 1e-1 / 1e-8.  Without equality constraints
    (l1l2) every order tried is as accurate as the greedy one."""
    has_eq = stages is not None and any(int(s) == 1 for s in stages)
    best = None
    for slack in ((None, 0) if has_eq else (None, 0, 1, 2, 3, 4, 6, 8, 12)):
        p = (min_degree_order(n, rows, cols, stages, "one") if slack is None
             else multiple_elimination_order(n, rows, cols, stages, slack))
        c = symbolic_ldl_cost(n, rows, cols, p)
        score = c["depth"] + c["flops"] / float(width)
        if best is None or score < best[0]:
            best = (score, p, slack, c)
    return best[1]


class CSparse:
    def __init__(self):
        self.table = InstructionsTable()
        self.cache = {}            # node idx -> (subs, ins)
        self.alias = {}            # variable node idx -> source node
        self.load_value = {}       # instruction id -> constant (I_load)
        self.sets, self.gets, self.copies = [], [], []
        self.ldl_perm = {}         # ldl node idx -> permutation (0-based)
        self.ldl_order = None      # None -> module attribute LDL_ORDER
        self.ldl_stats = []
        self.lu_perm = {}          # lu node idx -> (row permutation, column permutation)
        self.lu_stats = []
        self._one = None

    # ---- declare* API (csparse.m:296-650) --------------------------------------
    def declareSet(self, var, name, external=False):
        if var.op != "variable":
            raise ValueError("declareSet: destination must be a Tvariable")
        if var.idx in self.alias:
            raise ValueError("declareSet: cannot set an alias")
        self.scalarize(var)
        self.sets.append(dict(name=name, var=var, external=bool(external)))

    def declareGet(self, sources, name, output_names=None):
        if isinstance(sources, dict):
            output_names = list(sources.keys())
            sources = list(sources.values())
        if not isinstance(sources, (list, tuple)):
            sources = [sources]
        sources = [tc.toT(s) for s in sources]
        if not sources:
            raise ValueError("empty declareGet")
        for s in sources:
            self.scalarize(s)
        if output_names is None:
            output_names = [s.par if s.op == "variable" else f"output{k + 1}"
                            for k, s in enumerate(sources)]
        self.gets.append(dict(name=name, sources=sources, output_names=list(output_names)))

    def declareCopy(self, dests, sources, name):
        if not isinstance(dests, (list, tuple)):
            dests = [dests]
        if not isinstance(sources, (list, tuple)):
            sources = [sources]
        if len(dests) != len(sources):
            raise ValueError("declareCopy: mismatch between source/destination lists")
        d2, s2 = [], []
        for d, s in zip(dests, sources):
            s = tc.toT(s)
            if tc._prod(s.size) == 0:
                continue
            if d.op != "variable":
                raise ValueError("declareCopy: destination must be a Tvariable")
            if d.size != s.size:
                raise ValueError(f"declareCopy {name}: source size {s.size} differs from "
                                 f"destination size {d.size}")
            self.scalarize(d)
            self.scalarize(s)
            d2.append(d)
            s2.append(s)
        if not d2:
            raise ValueError("empty declareCopy")
        self.copies.append(dict(name=name, dests=d2, sources=s2))

    def declareAlias(self, source, name):
        """Name an expression so later expressions can refer to it as a variable whose
        scalar instructions *are* the source's (csparse.m:570-614)."""
        source = tc.toT(source)
        var = tc.Tvariable(name, source.size)
        if var.idx in self.cache or var.idx in self.alias:
            raise ValueError(f"declareAlias: linked variable '{name}' already existed")
        self.alias[var.idx] = source
        self.scalarize(var)
        return var

    # ---- instruction helpers -----------------------------------------------------
    def load(self, value):
        i = self.table.append(I.I_load, [float(value)], [])
        self.load_value[i] = float(value)
        return i

    def _all_ones(self, ins):
        lv = self.load_value
        return all(lv.get(int(i)) == 1.0 for i in np.unique(ins))

    def _append_grouped(self, typ, key_sorted, payload, params=None):
        """rows sorted by key; one instruction per run of equal keys whose operands are the
        concatenated payload rows. Returns (first_row_of_each_group, instr ids)."""
        start, cnt = _groups(key_sorted)
        out = np.zeros(start.size, dtype=np.int64)
        w = payload.shape[1]
        for c in np.unique(cnt):
            g = np.flatnonzero(cnt == c)
            rows = (start[g][:, None] + np.arange(c)[None, :]).reshape(-1)
            ops = payload[rows].reshape(g.size, c * w)
            par = None
            if params is not None:
                par = params(c, rows, g.size)
            out[g] = self.table.append_many(typ, par, ops)
        return start, cnt, out

    # ---- scalarization dispatch (computeScalarInstructions.m:165-433) ------------
    def scalarize(self, node):
        r = self.cache.get(node.idx)
        if r is not None:
            return r
        # iterative post-order to avoid deep recursion on long expression chains
        stack = [node]
        while stack:
            n = stack[-1]
            if n.idx in self.cache:
                stack.pop()
                continue
            deps = [self.alias[n.idx]] if n.idx in self.alias else list(n.args)
            pending = [a for a in deps if a.idx not in self.cache]
            if pending:
                stack.extend(pending)
                continue
            stack.pop()
            subs, ins = self._scalarize1(n)
            ins = np.ascontiguousarray(ins, dtype=np.int64).reshape(-1)
            subs = np.ascontiguousarray(subs, dtype=np.int64).reshape(ins.size, len(n.size))
            subs, ins = _sort_cm(subs, ins, n.size)
            self.cache[n.idx] = (subs, ins)
        return self.cache[node.idx]

    def _scalarize1(self, n):
        op = n.op
        if op == "variable":
            if n.idx in self.alias:
                return self.cache[self.alias[n.idx].idx]
            cnt = tc._prod(n.size)
            ins = self.table.append_many(I.I_set, None, np.zeros((cnt, 0), dtype=np.int64),
                                         unique=False)
            return _full_subs(n.size), ins
        if op == "zeros":
            return _empty(len(n.size))
        if op == "ones":
            one = self.load(1.0)
            return _full_subs(n.size), np.full(tc._prod(n.size), one, dtype=np.int64)
        if op == "eye":
            h = len(n.size) // 2
            s1 = _full_subs(n.size[:h])
            return np.hstack([s1, s1]), np.full(s1.shape[0], self.load(1.0), dtype=np.int64)
        if op == "constant":
            v = n.value
            lin = np.flatnonzero(v.reshape(-1, order="F"))
            vals = v.reshape(-1, order="F")[lin]
            ins = np.array([self.load(x) for x in vals], dtype=np.int64)
            return _unlin(lin, n.size), ins
        f = getattr(self, "_s_" + op, None)
        if f is None:
            raise NotImplementedError(f"csparse: no scalarizer for '{op}'")
        return f(n)

    def _arg(self, n, k=0):
        return self.cache[n.args[k].idx]

    # ---- aliasing (zero-instruction) operators -----------------------------------
    def _s_reshape(self, n):
        subs, ins = self._arg(n)
        return _unlin(_lin(subs, n.args[0].size), n.size), ins

    def _s_transpose(self, n):
        subs, ins = self._arg(n)
        return subs[:, ::-1], ins

    def _s_cat(self, n):
        d = n.par - 1
        off = 0
        S, Ins = [], []
        for a in n.args:
            subs, ins = self.cache[a.idx]
            s = subs.copy()
            s[:, d] += off
            off += a.size[d]
            S.append(s)
            Ins.append(ins)
        return np.vstack(S), np.concatenate(Ins)

    def _s_subsref(self, n):
        subs, ins = self._arg(n)
        for d, sel in enumerate(n.par):
            sel = np.asarray(sel, dtype=np.int64)
            if sel.size == n.args[0].size[d] and (sel == np.arange(sel.size)).all():
                continue
            rT, rS = _join(subs[:, d], sel)
            subs = subs[rT].copy()
            subs[:, d] = rS
            ins = ins[rT]
        return subs, ins

    def _s_diag(self, n):
        subs, ins = self._arg(n)
        if n.args[0].ndim == 1:
            return np.hstack([subs, subs]), ins
        k = subs[:, 0] == subs[:, 1]
        return subs[k, :1], ins[k]

    def _s_full(self, n):
        subs, ins = self._arg(n)
        tot = tc._prod(n.size)
        out = np.zeros(tot, dtype=np.int64)
        out[_lin(subs, n.size)] = ins
        if (out == 0).any():
            out[out == 0] = self.load(0.0)
        return _full_subs(n.size), out

    def _s_ispositive(self, n):
        return self._arg(n)

    _s_iszero = _s_ispositive

    # ---- plus (sparsity_plus.m:19-53) ----------------------------------------------
    def _s_plus(self, n):
        L, In, Sg = [], [], []
        for a, s in zip(n.args, n.par):
            subs, ins = self.cache[a.idx]
            L.append(_lin(subs, n.size))
            In.append(ins)
            Sg.append(np.full(ins.size, float(s)))
        lin = np.concatenate(L)
        ins = np.concatenate(In)
        sg = np.concatenate(Sg)
        k = np.lexsort((ins, lin))          # by entry, then by instruction id (normalize)
        lin, ins, sg = lin[k], ins[k], sg[k]
        start, cnt = _groups(lin)
        out = np.zeros(start.size, dtype=np.int64)
        single = (cnt == 1) & (sg[start] > 0)
        out[single] = ins[start[single]]
        for c in np.unique(cnt):
            g = np.flatnonzero((cnt == c) & ~single)
            if g.size == 0:
                continue
            rows = start[g][:, None] + np.arange(c)[None, :]
            out[g] = self.table.append_many(I.I_sum, sg[rows], ins[rows])
        return _unlin(lin[start], n.size), out

    # ---- tprod (sparsity_tprod.m:156-266) ----------------------------------------------
    def _s_tprod(self, n):
        inds, sums = n.par
        npos = len(n.size)
        nneg = len(sums)
        gsize = list(n.size) + list(sums)        # global index g: 0..npos-1 outputs, then sums

        def gid(i):
            return i - 1 if i > 0 else npos + (-i) - 1

        ops = []
        for a, ix in zip(n.args, inds):
            subs, ins = self.cache[a.idx]
            g = [gid(i) for i in ix]
            # repeated index inside one operand: keep the generalized diagonal
            keepcols = []
            for c, gg in enumerate(g):
                first = g.index(gg)
                if first != c:
                    m = subs[:, first] == subs[:, c]
                    subs, ins = subs[m], ins[m]
                else:
                    keepcols.append(c)
            subs = subs[:, keepcols]
            g = [g[c] for c in keepcols]
            ops.append(dict(g=g, subs=subs, ins=ins, ones=self._all_ones(ins) if ins.size else False))
        if any(o["ins"].size == 0 for o in ops):
            return _empty(npos)
        ng = npos + nneg
        # greedy join order: start from the smallest operand, prefer operands sharing indices
        remaining = list(range(len(ops)))
        first = min(remaining, key=lambda k: ops[k]["ins"].size)
        remaining.remove(first)
        o = ops[first]
        assign = np.full((o["ins"].size, ng), -1, dtype=np.int64)
        assign[:, o["g"]] = o["subs"]
        have = set(o["g"])
        fac = [] if o["ones"] else [o["ins"]]
        while remaining:
            k = max(remaining, key=lambda k: (len(have & set(ops[k]["g"])), -ops[k]["ins"].size))
            remaining.remove(k)
            o = ops[k]
            shared = [c for c, gg in enumerate(o["g"]) if gg in have]
            new = [c for c, gg in enumerate(o["g"]) if gg not in have]
            if shared:
                sz = [gsize[o["g"][c]] for c in shared]
                keyT = _lin(assign[:, [o["g"][c] for c in shared]], sz)
                keyS = _lin(o["subs"][:, shared], sz)
                rT, rS = _join(keyT, keyS)
            else:
                nT, nS = assign.shape[0], o["ins"].size
                rT = np.repeat(np.arange(nT, dtype=np.int64), nS)
                rS = np.tile(np.arange(nS, dtype=np.int64), nT)
            assign = assign[rT]
            for c in new:
                assign[:, o["g"][c]] = o["subs"][rS, c]
            fac = [f[rT] for f in fac]
            if not o["ones"]:
                fac.append(o["ins"][rS])
            have |= set(o["g"])
            if assign.shape[0] == 0:
                return _empty(npos)
        okey = _lin(assign[:, :npos], n.size)
        if not fac:
            # every operand was a 0/1 pattern: the entry is the number of matching terms
            k = np.argsort(okey, kind="stable")
            start, cnt = _groups(okey[k])
            ins = np.array([self.load(float(c)) for c in cnt], dtype=np.int64)
            return _unlin(okey[k][start], n.size), ins
        F = np.sort(np.stack(fac, axis=1), axis=1)            # sort factors in each product
        k = np.lexsort(tuple(F[:, c] for c in range(F.shape[1] - 1, -1, -1)) + (okey,))
        okey, F = okey[k], F[k]                               # sort the sums
        nf = F.shape[1]
        start, cnt = _groups(okey)
        out = np.zeros(start.size, dtype=np.int64)
        if nf == 1:
            single = cnt == 1
            out[single] = F[start[single], 0]
        else:
            single = np.zeros(start.size, dtype=bool)
        for c in np.unique(cnt):
            g = np.flatnonzero((cnt == c) & ~single)
            if g.size == 0:
                continue
            rows = (start[g][:, None] + np.arange(c)[None, :]).reshape(-1)
            opsm = F[rows].reshape(g.size, c * nf)
            par = np.tile(np.array([[float(nf), float(c)]]), (g.size, 1))
            out[g] = self.table.append_many(I.I_sumprod, par, opsm)
        return _unlin(okey[start], n.size), out

    # ---- element-wise / reductions -----------------------------------------------------
    def _s_rdivide(self, n):
        a, b = n.args
        sa, ia = self.cache[a.idx]
        sb, ib = self.cache[b.idx]
        if b.size == ():
            if ib.size == 0:
                raise ValueError("rdivide: structurally zero scalar denominator")
            subs = sa
            ops = np.stack([ia, np.full(ia.size, ib[0])], axis=1)
        elif a.size == ():
            if ia.size == 0:
                return _empty(len(n.size))
            if ib.size != tc._prod(b.size):
                raise ValueError("rdivide: structurally zero values in the denominator")
            subs = sb
            ops = np.stack([np.full(ib.size, ia[0]), ib], axis=1)
        else:
            if ib.size != tc._prod(b.size):
                raise ValueError("rdivide: structurally zero values in the denominator")
            subs = sa
            ops = np.stack([ia, ib[_lin(sa, a.size)]], axis=1)  # ib is full & column-major sorted
        if ops.shape[0] == 0:
            return _empty(len(n.size))
        return subs, self.table.append_many(I.I_div, None, ops)

    def _s_norm2(self, n):
        _, ins = self._arg(n)
        if ins.size == 0:
            return _empty(0)
        return np.zeros((1, 0), dtype=np.int64), [self.table.append(I.I_plus_sqr, [], np.sort(ins))]

    def _s_norm1(self, n):
        _, ins = self._arg(n)
        if ins.size == 0:
            return _empty(0)
        return np.zeros((1, 0), dtype=np.int64), [self.table.append(I.I_plus_abs, [], np.sort(ins))]

    def _s_norminf(self, n):
        _, ins = self._arg(n)
        if ins.size == 0:
            return _empty(0)
        typ = I.I_abs if ins.size == 1 else I.I_max_abs
        return np.zeros((1, 0), dtype=np.int64), [self.table.append(typ, [], np.sort(ins))]

    def _s_compose(self, n):
        subs, ins = self._arg(n)
        fun = n.par
        if fun == "2":
            two = self.load(2.0)
            return _full_subs(n.size), np.full(tc._prod(n.size), two, dtype=np.int64)
        if fun not in I.COMPOSE_KEEPS_ZERO:
            tot = tc._prod(n.size)
            allins = np.zeros(tot, dtype=np.int64)
            allins[_lin(subs, n.size)] = ins
            if (allins == 0).any():
                allins[allins == 0] = self.load(0.0)
            subs, ins = _full_subs(n.size), allins
        if ins.size == 0:
            return _empty(len(n.size))
        return subs, self.table.append_many(I.COMPOSE_OPCODE[fun], None, ins[:, None])

    def _s_componentwise(self, n):
        """sparsity_componentwise.m:35: one I_componentwise per entry (structurally full result); the parameters are
        the characters of the C text with 0 in place of the operand hole (instructionsTableUTHash.c:1333-1345)"""
        subs, ins = self._arg(n)
        tot = tc._prod(n.size)
        allins = np.zeros(tot, dtype=np.int64)
        allins[_lin(subs, n.size)] = ins
        if (allins == 0).any():
            allins[allins == 0] = self.load(0.0)
        a, b = n.par.split("%s")
        par = [float(ord(ch)) for ch in a] + [0.0] + [float(ord(ch)) for ch in b]
        return _full_subs(n.size), self.table.append_many(I.I_componentwise, np.tile(par, (tot, 1)), allins[:, None])

    def _minmax(self, n, t_full, t_zero):
        subs, ins = self._arg(n)
        x = n.args[0]
        dims = [d - 1 for d in n.par]
        keep = [d for d in range(x.ndim) if d not in dims]
        nred = tc._prod([x.size[d] for d in dims])
        okey = _lin(subs[:, keep], n.size)
        k = np.lexsort((ins, okey))
        okey, ins = okey[k], ins[k]
        start, cnt = _groups(okey)
        out = np.zeros(start.size, dtype=np.int64)
        for c in np.unique(cnt):
            g = np.flatnonzero(cnt == c)
            rows = start[g][:, None] + np.arange(c)[None, :]
            if c == nred:
                if c == 1:
                    out[g] = ins[rows[:, 0]]
                else:
                    out[g] = self.table.append_many(t_full, None, ins[rows])
            else:
                out[g] = self.table.append_many(t_zero, None, ins[rows])
        return _unlin(okey[start], n.size), out

    def _s_min(self, n):
        return self._minmax(n, I.I_min, I.I_min0)

    def _s_max(self, n):
        return self._minmax(n, I.I_max, I.I_max0)

    def _minmax2(self, n, t_full, t_zero):
        a, b = n.args
        sa, ia = self.cache[a.idx]
        sb, ib = self.cache[b.idx]
        lin = np.concatenate([_lin(sa, n.size), _lin(sb, n.size)])
        ins = np.concatenate([ia, ib])
        k = np.lexsort((ins, lin))
        lin, ins = lin[k], ins[k]
        start, cnt = _groups(lin)
        out = np.zeros(start.size, dtype=np.int64)
        g2 = np.flatnonzero(cnt == 2)
        if g2.size:
            out[g2] = self.table.append_many(t_full, None, np.stack([ins[start[g2]], ins[start[g2] + 1]], 1))
        g1 = np.flatnonzero(cnt == 1)
        if g1.size:
            out[g1] = self.table.append_many(t_zero, None, ins[start[g1]][:, None])
        return _unlin(lin[start], n.size), out

    def _s_min2(self, n):
        return self._minmax2(n, I.I_min, I.I_min0)

    def _s_max2(self, n):
        return self._minmax2(n, I.I_max, I.I_max0)

    def _s_clp(self, n):
        x, dx = n.args
        sx, ix = self.cache[x.idx]
        sd, idx_ = self.cache[dx.idx]
        if idx_.size == 0:
            raise ValueError("clp: structurally zero direction")
        lx = _lin(sx, x.size)
        ld = _lin(sd, x.size)
        pos = np.searchsorted(lx, ld)
        if (pos >= lx.size).any() or (lx[np.minimum(pos, lx.size - 1)] != ld).any():
            raise ValueError("clp: structurally zero x where dx is not structurally zero")
        pairs = np.stack([ix[pos], idx_], axis=1)
        k = np.lexsort((pairs[:, 1], pairs[:, 0]))               # sort inequalities (normalize)
        return np.zeros((1, 0), dtype=np.int64), [self.table.append(I.I_clp, [], pairs[k].reshape(-1))]

    # ---- LDL (sparsity_ldl.m:161-222) -----------------------------------------------------
    def _s_ldl(self, n):
        A = n.args[0]
        subs, ins = self.cache[A.idx]
        N = n.size[0]
        ent = {(int(r), int(c)): int(i) for (r, c), i in zip(subs, ins)}
        nonsym = sum(1 for (r, c), i in ent.items() if ent.get((c, r)) != i)
        tag, stages = n.par
        how = self.ldl_order or LDL_ORDER
        if how == "mindeg":
            p = min_degree_order(N, subs[:, 0], subs[:, 1], stages)
        elif isinstance(how, (tuple, list)) and how[0] == "depth":
            p = depth_aware_order(N, subs[:, 0], subs[:, 1], stages, width=int(how[1]))
        elif isinstance(how, (tuple, list)) and how[0] == "mmd":
            p = multiple_elimination_order(N, subs[:, 0], subs[:, 1], stages, slack=int(how[1]),
                                           eq_gate=how[2] if len(how) > 2 else "one")
        else:
            raise ValueError("ldl order: 'mindeg', ('depth', width) or ('mmd', slack[, 'one'|'two'|'allbut1'])")
        pinv = np.empty(N, dtype=np.int64)
        pinv[p] = np.arange(N)
        # permuted lower triangle (row>col) and diagonal: Xp(r,c)=A(p[r],p[c])
        lowcol = [dict() for _ in range(N)]   # col -> {row: instr}
        diagA = [0] * N
        for (r, c), i in ent.items():
            pr, pc = int(pinv[r]), int(pinv[c])
            if pr == pc:
                diagA[pr] = i
            elif pr > pc:
                lowcol[pc][pr] = i
        t = self.table
        Lrow = [dict() for _ in range(N)]     # row -> {col: instr}   (strictly lower)
        Lcol = [[] for _ in range(N)]         # col -> rows
        D = [0] * N
        n0 = len(t)
        for col in range(N):
            Lr = Lrow[col]
            iss = sorted(Lr)
            V = {}
            if iss:
                ops = np.array([[Lr[i], D[i]] for i in iss], dtype=np.int64)
                ops.sort(axis=1)
                v = t.append_many(I.I_sumprod, np.tile([[2.0, 1.0]], (len(iss), 1)), ops)
                V = dict(zip(iss, (int(x) for x in v)))
            if diagA[col]:
                if not iss:
                    D[col] = diagA[col]
                else:
                    D[col] = t.append(I.I_plus_minus_dot, [],
                                      [diagA[col]] + self._dot_ops([(Lr[i], V[i]) for i in iss]))
            else:
                if not iss:
                    raise ValueError("ldl needs pivoting: either use LU instead of LDL or make "
                                     "sure the diagonal is not zero")
                D[col] = t.append(I.I_minus_dot, [], self._dot_ops([(Lr[i], V[i]) for i in iss]))
            cand = set(lowcol[col])
            for i in iss:
                cand.update(r for r in Lcol[i] if r > col)
            for row in sorted(cand):
                Lrr = Lrow[row]
                isr = [i for i in iss if i in Lrr]
                a = lowcol[col].get(row, 0)
                dots = self._dot_ops([(Lrr[i], V[i]) for i in isr])
                if a:
                    if not isr:
                        ins_ = t.append(I.I_div, [], [a, D[col]])
                    else:
                        ins_ = t.append(I.I_plus_minus_dot_div, [], [a, D[col]] + dots)
                else:
                    ins_ = t.append(I.I_minus_dot_div, [], [D[col]] + dots)
                Lrr[col] = ins_
                Lcol[col].append(row)
        S, Ins = [], []
        for r in range(N):
            S.append((r, r))
            Ins.append(D[r])
            for c, i in Lrow[r].items():
                S.append((r, c))
                Ins.append(i)
                S.append((c, r))
                Ins.append(i)
        self.ldl_perm[n.idx] = p
        nnzL = sum(len(x) for x in Lrow)
        self.ldl_stats.append(dict(n=N, nnzA=len(ent), nonsym=nonsym, nnzL=nnzL,
                                   new_instructions=len(t) - n0, tag=tag))
        subs = np.array(S, dtype=np.int64).reshape(-1, 2)
        ins = np.array(Ins, dtype=np.int64)
        return _sort_cm(subs, ins, n.size)

    @staticmethod
    def _dot_ops(pairs):
        """[(a,b),...] -> flat operand list with factors and terms sorted (normalize)."""
        pr = sorted(tuple(sorted(p)) for p in pairs)
        return [x for p in pr for x in p]

    def _s_ldl_d(self, n):
        subs, ins = self._arg(n)
        k = subs[:, 0] == subs[:, 1]
        return subs[k, :1], ins[k]

    def _s_ldl_l(self, n):
        subs, ins = self._arg(n)
        k = subs[:, 0] > subs[:, 1]
        N = n.size[0]
        d = np.arange(N, dtype=np.int64)
        one = self.load(1.0)
        return (np.vstack([subs[k], np.stack([d, d], 1)]),
                np.concatenate([ins[k], np.full(N, one, dtype=np.int64)]))

    def _typical(self, tag, size, subs):
        """values of the matrix `tag` saved by an earlier run (<typical_folder>/<tag>.subscripts/.values, tenscalc_amd/typical.py)
        in the order of `subs`, or None (no files, another sparsity structure, mismatching magic numbers:
        sparsity_lu.m:38-41 then falls back to random values)"""
        folder = getattr(self, "typical_folder", None)
        if not folder or not tag:
            return None
        from . import typical
        r = typical.load(os.path.join(folder, tag + ".subscripts"), os.path.join(folder, tag + ".values"))
        if r is None:
            return None
        osize, tsubs, tvals = r
        if tuple(osize) != tuple(size) or tsubs.shape != subs.shape:
            return None
        key = {tuple(x): k for k, x in enumerate(tsubs.tolist())}
        try:
            idx = [key[tuple(x)] for x in subs.tolist()]
        except KeyError:
            return None
        return tvals[idx]

    # ---- LU with row / column orders fixed at code-generation time (@csparse/sparsity_lu.m) ----------
    def _s_lu(self, n):
        """Right-looking symbolic Gaussian elimination of A(p,q).  p, q come from a NUMERICAL sparse LU of the matrix
        filled with "typical" values -- the reference draws 1+.5*rand per distinct instruction (sparsity_lu.m:44-50,
        MT19937 seed 0) and calls MATLAB's lu(A,.01,'vector') (UMFPACK, :78); here the same random values go through
        SuperLU (scipy.sparse.linalg.splu, COLAMD, diagonal pivot threshold .01): another fill-reducing order of
        the same kind, so fill and rounding differ from a MATLAB-generated solver, the algorithm does not.  The
        symbolic pass then follows sparsity_lu.m:136-195 line by line: in column `col` the pivot is the FIRST remaining
        row with a structural nonzero (:141-147), every other such row gets a multiplier L = a/pivot (I_div) and its
        entries right of `col` are updated with I_plus_minus_dot (entry present) or I_minus_dot (fill-in)."""
        import scipy.sparse as sp
        from scipy.sparse.linalg import splu
        A = n.args[0]
        subs, ins = self.cache[A.idx]
        N = n.size[0]
        if len(ins) == 0:
            raise ValueError("lu: empty matrix")
        uniq, inv = np.unique(ins, return_inverse=True)
        vals = (1 + .5 * np.random.RandomState(0).random_sample(len(uniq)))[inv]
        typ = self._typical(n.par[0], n.size, subs)
        how = "random values"
        if typ is not None:                      # sparsity_lu.m:51-57: the saved matrix + noise that removes exact zeros
            if np.isnan(typ).any():
                raise ValueError("typical values include %d nan entries" % int(np.isnan(typ).sum()))
            nz = np.abs(typ[typ != 0])
            tol = (nz.min() if nz.size else 1.0) * np.finfo(float).eps ** 2
            vals = typ + tol * np.random.RandomState(0).standard_normal(typ.size)
            how = "typical values"
        Atyp = sp.csc_matrix((vals, (subs[:, 0], subs[:, 1])), shape=(N, N))
        f = splu(Atyp, permc_spec="COLAMD", diag_pivot_thresh=0.01)
        p = np.argsort(f.perm_r)              # A[p][:, q] = L U
        q = np.argsort(f.perm_c)
        pinv = np.empty(N, dtype=np.int64)
        qinv = np.empty(N, dtype=np.int64)
        pinv[p] = np.arange(N)
        qinv[q] = np.arange(N)
        rows = [dict() for _ in range(N)]      # permuted row -> {permuted col: instr}
        colrows = [set() for _ in range(N)]
        tv = [dict() for _ in range(N)]        # typical values alongside (zero-pivot check, :154-156)
        for (r, c), i, v in zip(subs, ins, vals):
            pr, pc = int(pinv[r]), int(qinv[c])
            rows[pr][pc] = int(i)
            tv[pr][pc] = float(v)
            colrows[pc].add(pr)
        t = self.table
        n0 = len(t)
        remaining = set(range(N))
        pp = []
        for col in range(N):
            cand = sorted(r for r in colrows[col] if r in remaining)
            if not cand:
                raise ValueError("sparsity_lu: no pivot in column %d" % (col + 1))
            pivot = cand[0]
            remaining.discard(pivot)
            pp.append(pivot)
            if tv[pivot][col] == 0:
                raise ValueError("lu: unexpected zero pivot")
            prow = rows[pivot]
            pcols = sorted(c for c in prow if c > col)
            for row in cand[1:]:
                rr = rows[row]
                tv[row][col] = tv[row][col] / tv[pivot][col]
                l_ = t.append(I.I_div, [], [rr[col], prow[col]])
                rr[col] = l_
                for c in pcols:
                    if c in rr:
                        tv[row][c] = tv[row][c] - tv[row][col] * tv[pivot][c]
                        rr[c] = t.append(I.I_plus_minus_dot, [], [rr[c]] + self._dot_ops([(l_, prow[c])]))
                    else:
                        tv[row][c] = -tv[row][col] * tv[pivot][c]
                        rr[c] = t.append(I.I_minus_dot, [], self._dot_ops([(l_, prow[c])]))
                        colrows[c].add(row)
        pp = np.array(pp, dtype=np.int64)
        S, Ins = [], []
        for k, r in enumerate(pp):              # row k of L+U is the row that pivoted in column k (:203-204)
            for c, i in rows[int(r)].items():
                S.append((k, c))
                Ins.append(i)
        self.lu_perm[n.idx] = (p[pp], q)
        self.lu_stats.append(dict(n=N, nnzA=len(ins), nnzLU=len(Ins), new_instructions=len(t) - n0, tag=n.par[0], order_from=how,
                                  changed_pivots=int((pp != np.arange(N)).sum())))
        return _sort_cm(np.array(S, dtype=np.int64).reshape(-1, 2), np.array(Ins, dtype=np.int64), n.size)

    def _mldivide_lu(self, n):
        """lu(A)\\b = U\\(L\\b(p)) un-permuted by q: unit-lower forward substitution (sparsity_mldivide_l1.m:73-101),
        backward substitution with the pivots on U's diagonal (sparsity_mldivide_u.m:46-71, :73-74)."""
        F, b = n.args
        subs, ins = self.cache[F.idx]
        p, q = self.lu_perm[F.idx]
        N = F.size[0]
        pinv = np.empty(N, dtype=np.int64)
        pinv[p] = np.arange(N)
        low = [dict() for _ in range(N)]
        up = [dict() for _ in range(N)]
        D = [0] * N
        for (r, c), i in zip(subs, ins):
            r, c, i = int(r), int(c), int(i)
            if r == c:
                D[r] = i
            elif r > c:
                low[r][c] = i
            else:
                up[r][c] = i
        sb, ib = self.cache[b.idx]
        m = 1 if b.ndim == 1 else b.size[1]
        cols = [dict() for _ in range(m)]
        for s_, i in zip(sb, ib):
            cc = 0 if b.ndim == 1 else int(s_[1])
            cols[cc][int(pinv[int(s_[0])])] = int(i)
        t = self.table
        S, Ins = [], []
        for cc, X in enumerate(cols):
            if not X:
                continue
            for row in range(1, N):
                ks = [k for k in low[row] if k in X]
                if not ks:
                    continue
                dots = self._dot_ops([(low[row][k], X[k]) for k in ks])
                if row in X:
                    X[row] = t.append(I.I_plus_minus_dot, [], [X[row]] + dots)
                else:
                    X[row] = t.append(I.I_minus_dot, [], dots)
            for row in range(N - 1, -1, -1):
                if not D[row]:
                    raise ValueError("mldivide_u: zero in the diagonal")
                ks = [k for k in up[row] if k in X]
                if row in X and ks:
                    X[row] = t.append(I.I_plus_minus_dot_div, [],
                                      [X[row], D[row]] + self._dot_ops([(up[row][k], X[k]) for k in ks]))
                elif ks:
                    X[row] = t.append(I.I_minus_dot_div, [], [D[row]] + self._dot_ops([(up[row][k], X[k]) for k in ks]))
                elif row in X:
                    X[row] = t.append(I.I_div, [], [X[row], D[row]])
            for row, i in X.items():
                S.append((int(q[row]),) if b.ndim == 1 else (int(q[row]), cc))
                Ins.append(i)
        if not S:
            return _empty(b.ndim)
        return np.array(S, dtype=np.int64).reshape(-1, b.ndim), np.array(Ins, dtype=np.int64)

    # ---- A\b = L'\(D\(L\b(p)))(pinv)  (csparse.m:906-957, sparsity_mldivide_*.m) ----------
    def _s_mldivide(self, n):
        F, b = n.args
        if F.op == "lu":
            return self._mldivide_lu(n)
        subs, ins = self.cache[F.idx]
        p = self.ldl_perm[F.idx]
        N = F.size[0]
        pinv = np.empty(N, dtype=np.int64)
        pinv[p] = np.arange(N)
        low = [dict() for _ in range(N)]      # row -> {col: instr}, strictly lower
        up = [dict() for _ in range(N)]       # row -> {col: instr}, strictly upper
        D = [0] * N
        for (r, c), i in zip(subs, ins):
            r, c, i = int(r), int(c), int(i)
            if r == c:
                D[r] = i
            elif r > c:
                low[r][c] = i
            else:
                up[r][c] = i
        sb, ib = self.cache[b.idx]
        m = 1 if b.ndim == 1 else b.size[1]
        cols = [dict() for _ in range(m)]     # per rhs column: permuted row -> instr
        for s, i in zip(sb, ib):
            cc = 0 if b.ndim == 1 else int(s[1])
            cols[cc][int(pinv[int(s[0])])] = int(i)
        t = self.table
        S, Ins = [], []
        for cc, X in enumerate(cols):
            if not X:
                continue
            # forward: rows 2..n   (sparsity_mldivide_l1.m:73-101)
            for row in range(1, N):
                ks = [k for k in low[row] if k in X]
                if not ks:
                    continue
                dots = self._dot_ops([(low[row][k], X[k]) for k in ks])
                if row in X:
                    X[row] = t.append(I.I_plus_minus_dot, [], [X[row]] + dots)
                else:
                    X[row] = t.append(I.I_minus_dot, [], dots)
            # diagonal (sparsity_mldivide_d.m:61-75)
            for row in range(N - 1, -1, -1):
                if row in X:
                    X[row] = t.append(I.I_div, [], [X[row], D[row]])
            # backward (sparsity_mldivide_u1.m:185-217)
            for row in range(N - 1, -1, -1):
                ks = [k for k in up[row] if k in X]
                if not ks:
                    continue
                dots = self._dot_ops([(up[row][k], X[k]) for k in ks])
                if row in X:
                    X[row] = t.append(I.I_plus_minus_dot, [], [X[row]] + dots)
                else:
                    X[row] = t.append(I.I_minus_dot, [], dots)
            for row, i in X.items():
                S.append((int(p[row]),) if b.ndim == 1 else (int(p[row]), cc))
                Ins.append(i)
        if not S:
            return _empty(b.ndim)
        return np.array(S, dtype=np.int64).reshape(-1, b.ndim), np.array(Ins, dtype=np.int64)

    # ---- compile: dependency groups + memory map (dependencyGroups.m:36-455) --------------
    def compile(self, defines=None):
        return Tape.build(self, defines or {})


class Tape:
    """Flat scalar tape + entry points + dependency groups: the unit every back-end consumes."""

    @staticmethod
    def build(cs, defines):
        T = Tape()
        types, opStart, ops, parStart, pars = cs.table.export()
        n = types.size
        T.types, T.opStart, T.ops, T.parStart, T.pars = types, opStart, ops, parStart, pars
        T.n = n
        T.defines = dict(defines)
        T.ldl_stats = list(cs.ldl_stats)

        def ins_of(node):
            subs, ins = cs.cache[node.idx]
            return subs, ins

        def full_ins(node, what):
            subs, ins = ins_of(node)
            if ins.size != tc._prod(node.size):
                if what == "getGrad__":
                    raise ValueError("some optimization variables do not affect the criteria nor "
                                     "the constraints (look for zeros in gradient)")
                raise ValueError(f"{what}: non-full tensor of size {node.size} "
                                 f"({ins.size}/{tc._prod(node.size)} nonzeros); use full()")
            return ins  # already column-major sorted

        # entry points ---------------------------------------------------------------
        gets, copies, sets = [], [], []
        for g in cs.gets:
            gets.append(dict(name=g["name"], outputs=[dict(name=nm, size=s.size, ins=full_ins(s, g["name"]))
                                                        for nm, s in zip(g["output_names"], g["sources"])]))
        for c in cs.copies:
            dst, src = [], []
            for d, s in zip(c["dests"], c["sources"]):
                sd, idd = ins_of(d)
                ss, iss = ins_of(s)
                if sd.shape != ss.shape or (sd != ss).any():
                    raise ValueError(f"copy '{c['name']}': source and destination have different "
                                     "sparsity structures, you may need full() in the source")
                dst.append(idd)
                src.append(iss)
            copies.append(dict(name=c["name"], dst=np.concatenate(dst), src=np.concatenate(src)))
        for s in cs.sets:
            sets.append(dict(name=s["name"], varname=s["var"].par, size=s["var"].size,
                             ins=full_ins(s["var"], s["name"]), external=s["external"]))
        return Tape.from_instructions(types, opStart, ops, parStart, pars, sets, gets, copies, defines,
                                      ldl_stats=list(cs.ldl_stats))

    @staticmethod
    def from_instructions(types, opStart, ops, parStart, pars, sets, gets, copies, defines, ldl_stats=(), merge=()):
        """Tape from a raw instruction table (CSR arrays, 1-based operand ids) and entry-point lists
        sets=[{name,varname,size,ins,external}], gets=[{name,outputs:[{name,size,ins}]}],
        copies=[{name,dst,src}]: computes dependency groups, their order, parents/children.

        merge: lists of instruction ids whose dependency groups are to be ONE group (Tape.merged): the group is needed by
        whatever needs any of them and invalidated by whatever invalidates any of them -- coarser than the reference's
        partition (dependencyGroups.m:111-116), never less safe: a group is only ever evaluated more often, on
        unchanged inputs."""
        T = Tape()
        types = np.asarray(types, dtype=np.int32)
        n = types.size
        T.types, T.opStart, T.ops, T.parStart, T.pars = types, np.asarray(opStart), np.asarray(ops), \
            np.asarray(parStart), np.asarray(pars, dtype=np.float64)
        opStart, ops = T.opStart, T.ops
        T.n = n
        T.defines = dict(defines)
        T.ldl_stats = list(ldl_stats)
        T.sets, T.gets, T.copies = [], [], []
        nE = 0
        out_cols, in_cols = [], []       # (bit, instr array)
        for g in gets:
            e = dict(name=g["name"], outputs=[dict(o, ins=np.asarray(o["ins"], dtype=np.int64)) for o in g["outputs"]],
                     bit=nE)
            out_cols.append((nE, np.concatenate([o["ins"] for o in e["outputs"]])))
            nE += 1
            T.gets.append(e)
        for c in copies:
            e = dict(name=c["name"], dst=np.asarray(c["dst"], dtype=np.int64),
                     src=np.asarray(c["src"], dtype=np.int64), bit=nE)
            if e["dst"].shape != e["src"].shape:
                raise ValueError(f"copy '{c['name']}': source/destination length mismatch")
            out_cols.append((nE, e["src"]))
            nE += 1
            T.copies.append(e)
        nOut = nE
        for s_ in sets:
            e = dict(name=s_["name"], varname=s_["varname"], size=tuple(s_["size"]),
                     ins=np.asarray(s_["ins"], dtype=np.int64), external=bool(s_["external"]), bit=nE)
            in_cols.append((nE, e["ins"]))
            nE += 1
            T.sets.append(e)
        for e in T.copies:
            e["dbit"] = nE
            in_cols.append((nE, e["dst"]))
            nE += 1
        T.nOutCols, T.nCols = nOut, nE

        # dependency signatures --------------------------------------------------------
        sig = [0] * (n + 1)
        for bit, ins in out_cols:
            for i in np.unique(ins):
                sig[int(i)] |= 1 << bit
        opl = ops.tolist()
        ost = opStart.tolist()
        for i in range(n, 0, -1):          # gets/copies that need i  (parentsOf)
            s = sig[i]
            if s:
                for j in opl[ost[i - 1]:ost[i]]:
                    sig[j] |= s
        outmask = (1 << nOut) - 1
        used = np.array([False] + [bool(sig[i] & outmask) for i in range(1, n + 1)])
        down = [0] * (n + 1)
        for bit, ins in in_cols:
            for i in np.unique(ins):
                down[int(i)] |= 1 << bit
        for i in range(1, n + 1):          # sets/copies that i depends on  (childrenOf)
            d = down[i]
            for j in opl[ost[i - 1]:ost[i]]:
                d |= down[j]
            down[i] = d
        isset = types == I.I_set
        isload = types == I.I_load
        # groups: unique (needed-by, depends-on) signatures of computed instructions
        keyof = {}
        group = np.zeros(n + 1, dtype=np.int64)
        forced = {}
        for q, members in enumerate(merge, 1):
            members = [int(i) for i in members if used[int(i)] and not isset[int(i) - 1] and not isload[int(i) - 1]]
            ks, kd = 0, 0
            for i in members:
                ks |= sig[i]
                kd |= down[i]
            for i in members:
                forced[i] = (ks, kd, q)
        for i in range(1, n + 1):
            if not used[i] or isset[i - 1] or isload[i - 1]:
                continue
            k = forced.get(i) or (sig[i], down[i])
            g = keyof.get(k)
            if g is None:
                g = len(keyof) + 1
                keyof[k] = g
            group[i] = g
        # topological order of groups: if g1 uses values of g2 then down(g1)>=down(g2) and
        # up(g2)>=up(g1) with one inclusion strict, so |down|-|up| strictly increases.
        keys = sorted(keyof, key=lambda k: (bin(k[1]).count("1") - bin(k[0]).count("1"), keyof[k]))
        if merge:
            # merged groups break the inclusion argument above: order by the dependencies themselves (Kahn, the
            # candidates taken in the order of the sort above), refusing a merge that closes a cycle
            import heapq
            rank = {keyof[k]: r for r, k in enumerate(keys)}
            preds = {g_: set() for g_ in rank}
            for i in range(1, n + 1):
                gi = int(group[i])
                if gi:
                    for j in opl[ost[i - 1]:ost[i]]:
                        gj = int(group[j])
                        if gj and gj != gi:
                            preds[gi].add(gj)
            succs = {g_: set() for g_ in rank}
            for g_, ps in preds.items():
                for p_ in ps:
                    succs[p_].add(g_)
            indeg = {g_: len(ps) for g_, ps in preds.items()}
            heap = [(rank[g_], g_) for g_, d_ in indeg.items() if d_ == 0]
            heapq.heapify(heap)
            order = []
            while heap:
                _, g_ = heapq.heappop(heap)
                order.append(g_)
                for s_ in succs[g_]:
                    indeg[s_] -= 1
                    if indeg[s_] == 0:
                        heapq.heappush(heap, (rank[s_], s_))
            if len(order) != len(rank):
                raise ValueError("Tape.merged: the merge closes a dependency cycle (a group between the merged ones "
                                 "must be merged with them)")
            byid = {keyof[k]: k for k in keys}
            keys = [byid[g_] for g_ in order]
        remap = {keyof[k]: r + 1 for r, k in enumerate(keys)}
        group = np.array([remap.get(int(g), 0) for g in group], dtype=np.int64)
        T.nGroups = len(keys)
        T.group = group                    # indexed by instruction id (1-based), 0 = none
        T.group_sig = [(k[0], k[1]) for k in keys]
        T.groups = [np.flatnonzero(group == g + 1) for g in range(T.nGroups)]  # ascending ids
        T.used = used
        for e in T.gets + T.copies:
            e["parents"] = [g + 1 for g, k in enumerate(keys) if k[0] >> e["bit"] & 1]
        for e in T.sets:
            e["children"] = [g + 1 for g, k in enumerate(keys) if k[1] >> e["bit"] & 1]
        for e in T.copies:
            e["children"] = [g + 1 for g, k in enumerate(keys) if k[1] >> e["dbit"] & 1]
        # verify topological validity
        for g, members in enumerate(T.groups):
            for i in members:
                for j in opl[ost[i - 1]:ost[i]]:
                    gj = group[j]
                    if gj > g + 1:
                        raise AssertionError("dependency groups are not topologically ordered")
        # memory map: one slot per live value (sets, constants, used computed instructions)
        slot = np.full(n + 1, -1, dtype=np.int64)
        live = np.flatnonzero(np.r_[False, used[1:] | isset])
        needed_sets = set()
        for _, ins in in_cols:
            needed_sets.update(int(i) for i in ins)
        k = 0
        for i in live:
            if isset[i - 1] and not (used[i] or i in needed_sets):
                continue
            slot[i] = k
            k += 1
        T.slot = slot
        T.nSlots = k
        T.isset, T.isload = isset, isload
        T.flops = T._count_flops()
        return T

    def merged(self, group_sets):
        """A tape with the same instructions and entry points whose dependency groups `group_sets` (lists of 1-based
        group numbers of THIS tape) are merged, each list into one group.  Used by the LDS-resident back-end to walk the
        KKT factorisation and the triangular solves as one level schedule (hip_emitter.emit_class_ldsm, fuse_chain)."""
        sets = [dict(name=e["name"], varname=e["varname"], size=e["size"], ins=e["ins"], external=e["external"]) for e in self.sets]
        gets = [dict(name=e["name"], outputs=[dict(name=o["name"], size=o["size"], ins=o["ins"]) for o in e["outputs"]])
                for e in self.gets]
        copies = [dict(name=e["name"], dst=e["dst"], src=e["src"]) for e in self.copies]
        merge = [np.concatenate([self.groups[g_ - 1] for g_ in gs]) for gs in group_sets]
        return Tape.from_instructions(self.types, self.opStart, self.ops, self.parStart, self.pars, sets, gets, copies,
                                      self.defines, ldl_stats=self.ldl_stats, merge=merge)

    # static flop-class counts per group: the accounting of writeCinstructionsC, case by case
    # (instructionsTableUTHash.c:816-832 for the classes, :962-1534 for the countFlops_* updates); the dynamic total of
    # a solve -- the sum over executed groups, writeCfunctionpergroup.m:481-487 -- is what examples/flops.m:29-42 tabulates
    def _count_flops(self):
        cnt = np.zeros((self.nGroups + 1, 17), dtype=np.int64)
        ty, os_ = self.types, self.opStart
        for i in range(1, self.n + 1):
            g = self.group[i]
            if g == 0:
                continue
            t = int(ty[i - 1])
            no = int(os_[i] - os_[i - 1])
            c = cnt[g]
            if t == I.I_sum:                                                   # :962
                c[I.P_nsum] += no - 1
            elif t == I.I_sumprod:                                             # :1004-1005
                m, s = self.pars[self.parStart[i - 1]:self.parStart[i]]
                c[I.P_nsum] += int(s) - 1
                c[I.P_nprod] += int(s) * (int(m) - 1)
            elif t in (I.I_plus_minus_dot_div, I.I_minus_dot_div):           # :1077-1079, :1102-1104
                c[I.P_nsum] += (no - 1) // 2
                c[I.P_nprod] += (no - 1) // 2
                c[I.P_ndiv] += 1
            elif t in (I.I_plus_minus_dot, I.I_minus_dot):                   # :1125-1126, :1148-1149: C integer division,
                c[I.P_nsum] += (no - 1) // 2                                   # so I_minus_dot (2k operands) counts k-1
                c[I.P_nprod] += (no - 1) // 2
            elif t in (I.I_div, I.I_inv):                                     # :1170, :1428
                c[I.P_ndiv] += 1
            elif t == I.I_plus_sqr:                                            # :1198-1199
                c[I.P_nsum] += no - 1
                c[I.P_nprod] += no
            elif t == I.I_plus_abs:                                            # :1214-1215
                c[I.P_nsum] += no - 1
                c[I.P_nabs] += no
            elif t in (I.I_min, I.I_max):                                     # :1229-1230, :1263-1264
                c[I.P_nsum] += no - 1
                c[I.P_nif] += no - 1
            elif t in (I.I_min0, I.I_max0):                                   # :1247-1248, :1281-1282
                c[I.P_nsum] += no
                c[I.P_nif] += no
            elif t == I.I_max_abs:                                             # :1297-1299
                c[I.P_nsum] += no - 1
                c[I.P_nif] += no - 1
                c[I.P_nabs] += no
            elif t == I.I_clp:                                                 # :1316
                c[I.P_nclp] += no
            elif t == I.I_round:
                c[I.P_nround] += 1
            elif t == I.I_ceil:
                c[I.P_nceil] += 1
            elif t == I.I_floor:
                c[I.P_nfloor] += 1
            elif t == I.I_abs:
                c[I.P_nabs] += 1
            elif t == I.I_sign:
                c[I.P_nsign] += 1
            elif t in (I.I_2, I.I_zero):                                      # :1378: no count
                pass
            elif t == I.I_sqrt:
                c[I.P_nsqrt] += 1
            elif t in (I.I_Dsqrt, I.I_DDsqrt):                                # :1416-1417, :1422-1423
                c[I.P_npow] += 1
                c[I.P_nprod] += 1
            elif t in (I.I_sqr, I.I_2times, I.I_6times):
                c[I.P_nprod] += 1
            elif t in (I.I_cube, I.I_3sqr):
                c[I.P_nprod] += 2
            elif t == I.I_minus_inv_sqr:
                c[I.P_nprod] += 1
                c[I.P_ndiv] += 1
            elif t == I.I_2_inv_cube:
                c[I.P_nprod] += 2
                c[I.P_ndiv] += 1
            elif t in (I.I_cos, I.I_sin, I.I_minus_cos, I.I_minus_sin, I.I_atan):
                c[I.P_ntrig] += 1
            elif t == I.I_log:
                c[I.P_nlog] += 1
            elif t == I.I_exp:
                c[I.P_nexp] += 1
            elif t == I.I_Datan:                                               # :1489-1491
                c[I.P_ndiv] += 1
                c[I.P_nsum] += 1
                c[I.P_nprod] += 1
            elif t == I.I_DDatan:                                              # :1497-1499
                c[I.P_ndiv] += 2
                c[I.P_nsum] += 2
                c[I.P_nprod] += 3
            elif t == I.I_srelu:
                c[I.P_nexp] += 1
                c[I.P_nlog] += 1
            elif t == I.I_dsrelu:
                c[I.P_nexp] += 1
                c[I.P_ndiv] += 1
            elif t == I.I_ddsrelu:
                c[I.P_nexp] += 2
                c[I.P_ndiv] += 1
                c[I.P_nsum] += 1
            elif t in (I.I_heaviside, I.I_relu):
                c[I.P_nif] += 1
            else:                                                              # I_componentwise :1334
                c[I.P_cpwise] += 1
        return cnt

    def reference_memory_map(self):
        """The reference's instruction -> scratchbook map (dependencyGroups.m:275-422), restated for comparison with
        `code.statistics.sizeScratchbook` (examples/flops.m:121): every instruction starts with a location of its own
        (sets and constants included, :277); inside each dependency group a *local* variable -- no reader outside the
        group, not the source of a get / copy (:316-319) -- takes, in instruction order, the first location of an
        earlier local variable whose last reader precedes it (:357-375), the location it gives up is never handed
        out again (:373); holes are then compressed (:405-422).  Returns (location per instruction, 1-based ids, 0-based
        locations, -1 for none; sizeScratchbook).  The lowerings here have memory maps of their own (lowering.py,
        ldsm_schedule.py); this one exists to be checked against the reference's number."""
        n = self.n
        ost, opl = self.opStart.tolist(), self.ops.tolist()
        grp = self.group
        last_user = np.zeros(n + 1, dtype=np.int64)
        outside = np.zeros(n + 1, dtype=bool)
        for i in range(1, n + 1):
            gi = grp[i]
            for j in opl[ost[i - 1]:ost[i]]:
                if i > last_user[j]:
                    last_user[j] = i
                if grp[j] != gi:
                    outside[j] = True
        outs = np.zeros(n + 1, dtype=bool)
        for e in self.gets:
            for o in e["outputs"]:
                outs[o["ins"]] = True
        for e in self.copies:
            outs[e["src"]] = True
        loc = np.arange(n + 1, dtype=np.int64)
        INF = np.iinfo(np.int64).max
        # unused instructions share the reference's all-zero dependency row, i.e. one more group (:111-116)
        members_of = [np.flatnonzero((grp == 0) & np.r_[False, ~self.isset])] + list(self.groups)
        for members in members_of:
            lv = [int(i) for i in members if not outside[i] and not outs[i]]
            m = len(lv)
            if m < 2:
                continue
            last = [int(last_user[i]) if last_user[i] > 0 else INF for i in lv]
            size = 1
            while size < m:
                size *= 2
            tree = [INF] * (2 * size)          # min segment tree over lastUsedMemory: "first i with value < v"
            for k in range(m):
                tree[size + k] = last[k]
            for k in range(size - 1, 0, -1):
                tree[k] = min(tree[2 * k], tree[2 * k + 1])

            def put(k, v):
                k += size
                tree[k] = v
                k //= 2
                while k:
                    tree[k] = min(tree[2 * k], tree[2 * k + 1])
                    k //= 2

            for j, v in enumerate(lv):
                if tree[1] >= v:
                    continue
                k = 1
                while k < size:
                    k = 2 * k if tree[2 * k] < v else 2 * k + 1
                i2 = k - size
                loc[v] = lv[i2]                 # localMemory(i): the location variable i2 started with
                put(i2, last[j])
                put(j, INF)
        _, inv = np.unique(loc[1:], return_inverse=True)
        out = np.r_[-1, inv].astype(np.int64)
        return out, int(inv.max()) + 1 if n else 0

    def flops_of_groups(self, groups):
        """nsum+nprod+ndiv+... for a list of 1-based group ids"""
        return int(self.flops[list(groups)].sum()) if groups else 0

    def instr(self, i):
        """(type, params, operands) of 1-based instruction i"""
        return (int(self.types[i - 1]), self.pars[self.parStart[i - 1]:self.parStart[i]],
                self.ops[self.opStart[i - 1]:self.opStart[i]])

    def entry(self, name):
        for kind, lst in (("set", self.sets), ("get", self.gets), ("copy", self.copies)):
            for k, e in enumerate(lst):
                if e["name"] == name:
                    return kind, k, e
        raise KeyError(name)

    def has(self, name):
        try:
            self.entry(name)
            return True
        except KeyError:
            return False

    def summary(self):
        return dict(instructions=int(self.n), used=int(self.used.sum()), groups=int(self.nGroups),
                    slots=int(self.nSlots), sets=len(self.sets), gets=len(self.gets),
                    copies=len(self.copies), ldl=self.ldl_stats)
