"""Tape analysis shared by the HIP lowerings: storage classes, expression inlining and affine
instruction families.

The reference's C emitter rolls >=50 consecutive same-shape instructions with constant operand
strides into a ``for`` loop (/root/reference/lib/csparse/instructionsTableUTHash.c:881-944,
``minInstructions4loop`` parameters4all.m:438-447).  We recover the same regularity from better
information: every vectorized tensor operation remembers the scalar instructions it produced
(``CSparse.cache``), so each tensor node is a candidate *family* -- same opcode/shape, operands in
arithmetic progression of storage index.  Two kinds are recognised:

* V1  element families: n instructions ``out[c] = f(leaf_0[b0+d0*c], leaf_1[b1+d1*c], ...)``;
* V2  term families inside one long ``I_sumprod``/``I_plus_sqr``: ``sum_k prod_p leaf_p[b+d*k]``.

Values used once inside their own dependency group are inlined into the consumer as nested
(parenthesised) expressions, which keeps the reference's evaluation order exactly.
"""
from __future__ import annotations

import os

import numpy as np

from . import instructions as I

MAX_TERMS_PERLINE = 100   # instructionsTableUTHash.c:173
V1_MIN = 6                # smallest element family rolled into a loop
MINE_MIN = 4              # smallest mined family
V2_MIN = 12               # smallest term family rolled into a loop


class Leaf:
    __slots__ = ("ins", "kind", "idx", "val")

    def __init__(self, ins, kind, idx, val=None):
        self.ins, self.kind, self.idx, self.val = ins, kind, idx, val


class Node:
    __slots__ = ("ins", "typ", "par", "kids")

    def __init__(self, ins, typ, par, kids):
        self.ins, self.typ, self.par, self.kids = ins, typ, par, kids


class Family:
    """V1 element family: members (instruction ids) share one template tree; leaf j of member c is
    at index base[j] + delta[j]*c; destination private slot dst0 + ddst*c."""

    def __init__(self, members, template, base, delta, dst0, ddst, var="c_"):
        self.members, self.template = members, template
        self.base, self.delta, self.dst0, self.ddst = base, delta, dst0, ddst
        self.n = len(members)
        self.var = var              # name of the loop index in the printed code


class DFamily:
    """Element family whose members are DISTRIBUTED over the P lanes that share one problem
    instance: member c is owned by lane c % P and lives in element c // P of a per-lane register
    array.  leaves: per flattened template leaf, either ("d", vec) for an aligned distributed
    operand (element c of vector `vec`) or ("r", Leaf) for a replicated operand."""

    def __init__(self, fid, members, template, leaves):
        self.fid, self.members, self.template, self.leaves = fid, members, template, leaves
        self.n = len(members)


class Lowering:
    # slot sharing: exempt volatile values from surviving a whole inner loop (see life()).  Sound (CPU replay of
    # real call traces, bit-exact on the GPU) and worth 26 slots on sls -- a seventh wavefront per CU -- but the
    # tighter reuse costs as much as the wavefront gains (0.72 vs 0.66 ms at equal occupancy, 0.66 at 7 per CU):
    # off by default, TC_FINE_LIVES=1 to enable (read when a Lowering is made; hashed into the class build key)
    FINE_LIVES = None

    def __init__(self, tape, code=None, roll=True, lanes=1, _replicate=(), mine=None, spill_all=False,
                 share=None):
        if self.FINE_LIVES is None:
            self.FINE_LIVES = os.environ.get("TC_FINE_LIVES", "0") == "1"
        T = self.T = tape
        # share: the solver #define block -> values of transient groups whose life times inside one pass of
        # the Newton loop do not overlap share storage (_assign_shared_slots); None: one slot per value
        self.share = share
        self.P = int(lanes)
        # spill_all: every computed value gets a memory slot (no inlining, no SSA temporaries): the
        # storage map of the table-driven back-end (interp_emitter), where instructions are data
        self.spill_all = bool(spill_all)
        self._replicate = set(_replicate)    # inputs that must not be distributed (decided below)
        # "mined" families: same-shape, mutually independent instructions found by scanning each group
        # (the reference's run detection, instructionsTableUTHash.c:881-944, without needing the
        # members to be consecutive); on by default only when lanes share an instance, where a family
        # is work split over the lanes -- with one thread per instance it would only move values
        # from registers to memory
        self.mine = (lanes > 1) if mine is None else bool(mine)
        n = T.n
        opl, ost = T.ops.tolist(), T.opStart.tolist()
        self._opl, self._ost = opl, ost
        grp = T.group
        # ---- external inputs (parameters / initial values): SoA block S --------------------
        self.external = np.zeros(n + 1, dtype=bool)
        self.ext_index = np.full(n + 1, -1, dtype=np.int64)
        self.inputs = []
        off = 0
        doff = 0
        self.dval = {}              # instruction id -> (vector key, element index)  [distributed values]
        self.dvec_len = {}          # vector key -> number of elements
        P = self.P
        for e in T.sets:
            if e["external"]:
                ins = e["ins"]
                sz = e["size"]
                n1 = sz[0] if len(sz) else 1
                dist = (P > 1 and len(sz) >= 1 and n1 >= 2 * P and len(ins) >= 4 * P
                        and e["varname"] not in self._replicate)
                self.external[ins] = True
                d = dict(name=e["varname"], entry=e["name"], size=sz, numel=len(ins), dist=dist)
                if dist:
                    # first dimension spread over the lanes: element (row, r) -> lane row % P,
                    # per-lane slot r*K + row // P; rows padded to K*P with zeros
                    K = -(-n1 // P)
                    rest = len(ins) // n1
                    d.update(n1=n1, K=K, rest=rest, off=doff, arr=len([x for x in self.inputs if x["dist"]]))
                    for q, i in enumerate(ins):
                        key = ("X", d["arr"], q // n1)
                        self.dval[int(i)] = (key, q % n1)
                        self.dvec_len[key] = n1
                    doff += rest * K
                else:
                    self.ext_index[ins] = off + np.arange(len(ins))
                    d.update(off=off)
                    off += len(ins)
                self.inputs.append(d)
        self.nExt = off             # replicated inputs: doubles per instance in the SoA block S
        self.nDist = doff           # distributed inputs: doubles per LANE in the block Sd
        out_needed = np.zeros(n + 1, dtype=bool)   # read by a get / copy source
        for e in T.gets:
            for o in e["outputs"]:
                out_needed[o["ins"]] = True
        for e in T.copies:
            out_needed[e["src"]] = True
        self.out_needed = out_needed
        # ---- readers -------------------------------------------------------------------------
        nread = np.zeros(n + 1, dtype=np.int64)       # in-tape readers (used instructions only)
        read_other = np.zeros(n + 1, dtype=bool)      # read from another group
        for i in range(1, n + 1):
            if not T.used[i] or grp[i] == 0:
                continue
            seen = set()
            for j in opl[ost[i - 1]:ost[i]]:
                if j in seen:
                    nread[j] += 1       # same consumer twice still counts (x*x)
                    continue
                seen.add(j)
                nread[j] += 1
                if grp[j] != grp[i]:
                    read_other[j] = True
        self.nread = nread
        computed = np.zeros(n + 1, dtype=bool)
        computed[1:] = (grp[1:] > 0)
        local = computed & ~read_other & ~out_needed
        # inline: local value with exactly one reader (one operand slot)
        inline = local & (nread == 1)
        self.single_local = inline.copy()    # one reader, in the same group, never read by an entry point (interp pair records)
        if spill_all:
            local = np.zeros(n + 1, dtype=bool)
            inline = np.zeros(n + 1, dtype=bool)
        self.computed = computed
        self.inline = inline
        self.local = local & ~inline                     # SSA temporaries
        self.promote = np.zeros(n + 1, dtype=bool)       # temporaries forced into memory
        # ---- candidate families from tensor nodes ---------------------------------------------------
        self.families = {}          # first member id -> Family
        self.in_family = np.zeros(n + 1, dtype=bool)
        cands = []
        if roll and code is not None:
            seen = set()
            for _, (subs, ins) in code.cache.items():
                if ins.size < V1_MIN:
                    continue
                key = (int(ins[0]), int(ins[-1]), ins.size)
                if key in seen:
                    continue
                seen.add(key)
                cands.append(ins)
        if P > 1:
            # an element-wise tensor node whose only reader is a reduction (y = A*x-b read by norm2 alone,
            # once the gradient no longer needs it) would be inlined into that reduction term by term;
            # keep it a family so that it can be distributed over the lanes with its operands
            long_sum = {}        # operand -> the long I_plus_sqr / I_sumprod that reads it
            for i in range(1, n + 1):
                if T.used[i] and grp[i] > 0 and T.types[i - 1] in (I.I_plus_sqr, I.I_sumprod) \
                        and ost[i] - ost[i - 1] >= 2 * P:
                    for j in opl[ost[i - 1]:ost[i]]:
                        long_sum[j] = i
            for ins in cands:
                if (ins.size >= 2 * P and self.inline[ins].all()
                        and len({long_sum.get(int(i), -int(i)) for i in ins}) == 1 and int(ins[0]) in long_sum
                        and all(any(int(j) in self.dval for j in opl[ost[int(i) - 1]:ost[int(i)]]) for i in ins)):
                    self.inline[ins] = False
                    self.local[ins] = True
        fam_members = []
        for ins in cands:
            m = self._family_precheck(ins)
            if m is not None:
                fam_members.append(m)
        self.dfamilies = {}         # first member id -> DFamily
        self.dreduce = {}           # sumprod/plus_sqr instruction id -> (nf, ns, factors)
        if P > 1:
            fam_members = self._classify_distributed(fam_members)
        fam_members = [self._best_order(m) for m in fam_members]
        mined = self._mine_families(fam_members) if (roll and self.mine) else []
        # members (and their strided computed leaves) must live in indexable memory
        for m in fam_members + mined:
            for i in m:
                if self.local[i]:
                    self.promote[i] = True
        self.local &= ~self.promote
        # ---- private slots: families first (contiguous), then everything else ---------------------------
        read_any = np.zeros(n + 1, dtype=bool)
        for i in range(1, n + 1):
            if T.used[i]:
                for j in opl[ost[i - 1]:ost[i]]:
                    read_any[j] = True
        read_any |= out_needed
        pslot = np.full(n + 1, -1, dtype=np.int64)
        blocks = []                      # storage requests: lists of instruction ids that need consecutive slots
        seen = np.zeros(n + 1, dtype=bool)
        for m in fam_members + mined:
            blk = [int(i) for i in m if not seen[i] and not self.inline[i]]
            if blk:
                seen[blk] = True
                blocks.append(blk)
        for i in range(1, n + 1):
            if seen[i]:
                continue
            if T.isset[i - 1]:
                if not self.external[i] and read_any[i]:
                    blocks.append([i])
            elif computed[i] and not self.local[i] and not self.inline[i] and i not in self.dval:
                blocks.append([i])
        if self.share is not None and not spill_all:
            k = self._assign_shared_slots(blocks, pslot)
        else:
            k = 0
            for blk in blocks:
                for i in blk:
                    pslot[i] = k
                    k += 1
        self.pslot = pslot
        self.nPriv = k
        # an input stays distributed only if every reader consumes it aligned (member c of a
        # distributed family or term c of a reduction); otherwise redo the analysis with it replicated
        if P > 1:
            bad = set()
            dm = {}
            for fid, m in getattr(self, "_dcand", {}).items():
                for c, i in enumerate(m):
                    dm[i] = c
            readers = {}
            for i in range(1, n + 1):
                if T.used[i] and grp[i] > 0:
                    for j in set(self.operands(i)):
                        readers.setdefault(j, []).append(i)
            for d in self.inputs:
                if not d["dist"]:
                    continue
                e = [x for x in T.sets if x["name"] == d["entry"]][0]
                for q, i in enumerate(e["ins"]):
                    i = int(i)
                    row = q % d["n1"]
                    if out_needed[i]:
                        bad.add(d["name"])
                        break
                    for r_ in readers.get(i, []):
                        top = self._consumer_root(r_, readers)
                        if dm.get(top) == row or dm.get(r_) == row or r_ in self.dreduce:
                            continue
                        bad.add(d["name"])
                        break
                    if d["name"] in bad:
                        break
            if bad:
                self.__init__(tape, code, roll, lanes, _replicate=self._replicate | bad, mine=self.mine,
                              share=self.share)
                return
        # ---- verify families against the final storage -----------------------------------------------------
        self._tree_cache = {}
        if P > 1:
            self.build_dfamilies()
        # every candidate (tensor-node or mined): keep its maximal affine sub-runs
        self.mined_pieces = {}                # first member of a candidate -> [Family | instruction id, ...]
        for m in fam_members + mined:
            pieces, k0 = [], 0
            while k0 < len(m):
                best = None
                if not self.in_family[m[k0]]:
                    for k1 in range(k0 + MINE_MIN, len(m) + 1):
                        if self.in_family[m[k0:k1]].any():
                            break
                        f = self._family_verify(m[k0:k1])
                        if f is None:
                            break
                        best = (k1, f)
                if best is None:
                    pieces.append(m[k0])
                    k0 += 1
                else:
                    pieces.append(best[1])
                    self.families[min(best[1].members)] = best[1]
                    self.in_family[m[k0:best[0]]] = True
                    k0 = best[0]
            self.mined_pieces[m[0]] = pieces
        self._build_units(mined)

    # -------------------------------------------------------------------------------------------
    def operands(self, i):
        return self._opl[self._ost[i - 1]:self._ost[i]]

    # ---- slot sharing ---------------------------------------------------------------------------------
    def _assign_shared_slots(self, blocks, pslot):
        """Values of transient dependency groups (certainly dirty at the head of the Newton loop) live from
        the first call that may evaluate their group to the last call that may evaluate a reader (or reads
        them itself); two values whose life times inside one pass of the loop body are disjoint get the
        same slot.  Everything else keeps a slot of its own: sets, copy destinations, groups that can stay
        clean across iterations, and every loop-evaluated group that getOutputs -- which runs after the
        loop, wherever the loop was left, and trusts the status flags -- depends on.

        Why this is sound with the lazy-evaluation flags left as they are: (1) inside the loop a value is
        read only while a reader group is being evaluated or by an entry point directly, and every
        position where that may happen is inside the value's life time (the sets E come from the same
        analysis that decides where groups are instantiated, and an inner loop counts as a whole);
        (2) a flag can say "clean" for a group whose storage a later group has reused only after that
        group's last possible reader, and before the next pass every transient group is invalidated;
        (3) after the loop only getOutputs runs, and nothing it depends on shares storage unless it is
        evaluated after the loop only.  The probe kernel (one get from scratch, all ancestors evaluated in
        one go) has no such schedule: classes built with sharing do not support tc_probe."""
        from .loop_schedule import timeline
        T = self.T
        tl = timeline(T, self.share)
        E, transient, loops, T_end = tl["E"], tl["transient"], tl["loops"], tl["T_end"]
        grp = T.group
        readers = {}
        for i in range(1, T.n + 1):
            if T.used[i] and grp[i] > 0:
                for j in set(self._opl[self._ost[i - 1]:self._ost[i]]):
                    readers.setdefault(j, []).append(i)
        direct = {}                          # instruction -> times at which an entry point reads it itself
        keep = set()                         # groups that must keep storage of their own
        for e in list(T.gets) + list(T.copies):
            ts = tl["time"].get(e["name"])
            if ts is None:                   # not called by the loop
                ts = [T_end]
                if e["name"].endswith("_getOutputs"):
                    keep |= {g_ for g_ in e["parents"] if E.get(g_)}
                else:                            # a get only the probe kernel would call: classes built with
                    continue                     # sharing have none (hip_emitter: no probes), nothing reads it
            for j in list(e.get("src", [])) + [x for o in e.get("outputs", []) for x in o["ins"]]:
                direct.setdefault(int(j), []).extend(ts)

        # groups that use lane-distributed state are PUSHED (hip_emitter: re-evaluated, with their ancestors, by the
        # set/copy that invalidates them), i.e. earlier than the first get that needs them: the life times
        # computed from the gets do not hold for them, they keep storage of their own
        if self.P > 1:
            dgroups = {int(grp[i]) for i in list(self.dval) + list(self.dreduce) if int(grp[i]) > 0}
            in_loop = set()
            for e in list(T.gets) + list(T.copies):
                if e["name"] in tl["time"]:
                    in_loop |= set(e["parents"])
            gdeps = {}
            for i in range(1, T.n + 1):
                if T.used[i] and grp[i] > 0:
                    for j in self._opl[self._ost[i - 1]:self._ost[i]]:
                        if grp[j] > 0 and grp[j] != grp[i]:
                            gdeps.setdefault(int(grp[i]), set()).add(int(grp[j]))
            todo = [g_ for g_ in dgroups if g_ in in_loop]
            while todo:
                g_ = todo.pop()
                if g_ not in keep:
                    keep.add(g_)
                    todo += list(gdeps.get(g_, ()))

        # groups no entry point of the solver ever invalidates (they depend on the parameters only): evaluated
        # once, in the first pass; a value of theirs that only other such groups read is dead after that
        varying = set()
        for e in list(T.sets) + list(T.copies):
            if e["name"] in tl["time"]:
                varying |= set(e["children"])

        def once(i):
            g = int(grp[i])
            return (g > 0 and g not in varying and g not in keep and i not in direct and E.get(g)
                    and all(int(grp[r]) not in varying and int(grp[r]) not in keep and E.get(int(grp[r]))
                            for r in readers.get(i, [])))

        # positions inside a call: the groups an entry point may evaluate run in the order of its parents list,
        # the entry point's own reads come last
        Epos = {}
        for e in list(T.gets) + list(T.copies):
            for t in tl["time"].get(e["name"], []):
                for k, g_ in enumerate(e["parents"]):
                    if g_ not in tl["clean"].get(e["name"], ()):
                        Epos.setdefault(g_, []).append(t + (k + 1) / 4096.0 if t >= 0 else -1)
        for j in direct:
            direct[j] = [t + 0.9999 if 0 <= t < T_end else t for t in direct[j]]
        # an inner loop repeats: what a later pass may read WITHOUT re-evaluating it must survive the whole loop.
        # Exempt are values of groups that every invalidating set inside the loop invalidates ("volatile"),
        # unless an entry point inside the loop reads them itself.  A further pass either follows such a set --
        # then every volatile group is dirty and is re-evaluated, in order, before anything reads it, and
        # non-volatile groups (clean) are neither evaluated nor do they read -- or it follows sets without
        # dependants only -- then nothing is evaluated and nothing is written, and only the entry points' own
        # reads happen again (those values are not exempt).  E.g. the LDL factor is dead once the triangular
        # solves have run and shares its slots with what the direction-error test computes next.
        volatile = []
        for p_, q_ in loops:
            inside = [e for e in list(T.sets) + list(T.copies)
                      if e["children"] and any(p_ <= t <= q_ for t in tl["time"].get(e["name"], []))]
            v = set.intersection(*[set(e["children"]) for e in inside]) if inside else set()
            volatile.append(v)

        def exempt(i, g, k):
            p_, q_ = loops[k]
            return g in volatile[k] and not any(p_ <= t < q_ + 1 for t in direct.get(i, []))

        # groups that no entry point the class ever calls depends on (gets that exist for the probe kernel only:
        # getGrad__, getJ__, getHessInertia__ when useInertia=0, ...).  Classes built with sharing have no probe
        # kernel, so a reader in such a group never runs; opt-in (ignore_dead_readers) because it changes the
        # packing of classes whose occupancy was tuned with the conservative rule
        dead = set()
        if getattr(self, "ignore_dead_readers", False):
            alive = set()
            for e in list(T.gets) + list(T.copies):
                if e["name"] in tl["time"] or e["name"].endswith("_getOutputs"):
                    alive |= set(e["parents"])
            dead = set(range(1, T.nGroups + 1)) - alive
        self.dead_groups = set(dead)         # diagnostics / ldsm_schedule.validate: ranges nothing ever walks

        def life(i):
            g = int(grp[i])
            if g == 0 or g in keep or (g not in transient and not once(i)):
                return None
            ts = list(Epos.get(g, [T_end]))
            for r in readers.get(i, []):
                if int(grp[r]) in dead:
                    continue
                ts += Epos.get(int(grp[r]), [T_end])
            ts += direct.get(i, [])
            # the initialisation sequence (position -1) is an epoch of its own: what it computes and the loop
            # recomputes before use is dead in between
            init = -1 in ts
            ts = [t for t in ts if t >= 0]
            if not ts:
                return init, T_end + 1, T_end + 1
            a, b = min(ts), max(ts)
            for k, (p_, q_) in enumerate(loops):
                if a < q_ + 1 and b >= p_ and not (self.FINE_LIVES and exempt(i, g, k)):
                    if self.FINE_LIVES and g in volatile[k] and a >= p_:
                        # volatile, but an entry point inside the loop reads it: a further pass without any
                        # evaluation reads it again, so it must last to the end of the loop -- but nothing can
                        # overwrite it between the loop head and its own position in such a pass (nothing is
                        # written at all), and in a pass that does evaluate it is recomputed first
                        b = max(b, q_ + 0.99999)
                    else:
                        a, b = min(a, p_), max(b, q_ + 0.99999)
            return init, a, b

        INF = 10 ** 9
        req = []                             # (block, used by the initialisation, a, b)
        for blk in blocks:
            lives = [life(i) for i in blk]
            if any(x is None for x in lives):
                req.append((blk, True, -INF, INF))
            else:
                req.append((blk, any(x[0] for x in lives), min(x[1] for x in lives), max(x[2] for x in lives)))

        self._last_req = req                 # diagnostics (scripts, tests): (block, init, first, last) per block
        if getattr(self, "_req_only", False):    # diagnostics: life times wanted, no packing
            return 0

        def pack(order):
            placed, where_of, size = [], {}, 0           # placed: (first slot, size, init, a, b)
            for n_ in order:
                blk, init, a, b = req[n_]
                nb = len(blk)
                where = None
                if a > -INF:
                    for s0 in sorted({0} | {p0 + sz for p0, sz, *_ in placed}):
                        if all(not (p0 < s0 + nb and s0 < p0 + sz) or ((b < pa or pb < a) and not (init and pi))
                               for p0, sz, pi, pa, pb in placed):
                            where = s0
                            break
                if where is None:
                    where = size
                where_of[n_] = where
                placed.append((where, nb, init, a, b))
                size = max(size, where + nb)
            return size, where_of
        # first fit is order dependent: try a few orders (as given; persistent first then by birth; by size) and
        # keep the tightest packing
        idx = list(range(len(req)))
        orders = [idx,
                  sorted(idx, key=lambda n_: (req[n_][2] > -INF, req[n_][2], -len(req[n_][0]))),
                  sorted(idx, key=lambda n_: (req[n_][2] > -INF, -len(req[n_][0]), req[n_][2])),
                  sorted(idx, key=lambda n_: (req[n_][2] > -INF, -(req[n_][3] - req[n_][2]), -len(req[n_][0])))]
        size, where_of = min((pack(o) for o in orders), key=lambda r: r[0])
        for n_, (blk, *_rest) in enumerate(req):
            for q, i in enumerate(blk):
                pslot[i] = where_of[n_] + q
        return size

    def leaf(self, j):
        T = self.T
        j = int(j)
        if T.isload[j - 1]:
            return Leaf(j, "lit", -1, float(T.pars[T.parStart[j - 1]]))
        if j in self.dval:
            return Leaf(j, "dval", self.dval[j])
        if self.external[j]:
            return Leaf(j, "ext", int(self.ext_index[j]))
        if self.local[j]:
            return Leaf(j, "tmp", j)
        if self.pslot[j] < 0:
            raise AssertionError(f"instruction {j} has no storage")
        return Leaf(j, "priv", int(self.pslot[j]))

    def tree(self, i):
        """expression tree of instruction i with single-use local operands inlined"""
        t = self._tree_cache.get(i)
        if t is None:
            typ, par, ops = self.T.instr(i)
            kids = [self.tree(int(o)) if self.inline[o] else self.leaf(o) for o in ops]
            t = Node(i, typ, tuple(float(p) for p in par), kids)
            self._tree_cache[i] = t
        return t

    def shape_sig(self, i, depth=0):
        """structure signature that ignores storage (used before slots exist)"""
        typ, par, ops = self.T.instr(i)
        kids = []
        for o in ops:
            o = int(o)
            if self.inline[o]:
                kids.append(self.shape_sig(o, depth + 1))
            elif self.T.isload[o - 1]:
                kids.append(("lit", float(self.T.pars[self.T.parStart[o - 1]])))
            else:
                kids.append("v")
        return (typ, tuple(float(p) for p in par), tuple(kids))

    def _family_precheck(self, ins):
        T = self.T
        ins = [int(i) for i in ins]
        if len(set(ins)) != len(ins):
            return None
        g = T.group[ins[0]]
        if g == 0:
            return None
        for i in ins:
            if T.group[i] != g or self.inline[i] or not self.computed[i]:
                return None
        s0 = self.shape_sig(ins[0])
        for i in ins[1:]:
            if self.shape_sig(i) != s0:
                return None
        # every leaf operand must exist before the first member is evaluated
        first = min(ins)
        mem = set(ins)
        for i in ins:
            for j in self._leaf_ids(i):
                if j in mem:
                    return None
                if j > first and T.group[j] == g:
                    return None
        return ins

    # ---- distributed (wave-cooperative) classification --------------------------------------------
    def _reduction_shape(self, q, dfam_of):
        """If instruction q (I_sumprod / I_plus_sqr) sums over aligned distributed operands return
        (nf, ns, [("d", vec) | ("r", instr id)] per factor), else None."""
        T = self.T
        typ, par, ops = T.instr(q)
        if typ == I.I_sumprod:
            nf, ns = int(par[0]), int(par[1])
        elif typ == I.I_plus_sqr:
            nf, ns = 1, len(ops)
        else:
            return None
        if ns < 2 * self.P:
            return None
        facs = []
        anyd = False
        for p in range(nf):
            o0 = int(ops[p])
            if o0 in self.dval and (self.dval[o0][0][0] == "X" or self.dval[o0][0] in dfam_of):
                vec = self.dval[o0][0]
                for k in range(ns):
                    o = int(ops[k * nf + p])
                    if self.dval.get(o) != (vec, k):
                        return None
                facs.append(("d", vec))
                anyd = True
            else:
                for k in range(ns):
                    o = int(ops[k * nf + p])
                    if o in self.dval:
                        return None
                    if o != o0:
                        if not (T.isload[o - 1] and T.isload[o0 - 1]
                                and T.pars[T.parStart[o - 1]] == T.pars[T.parStart[o0 - 1]]):
                            return None
                facs.append(("r", o0))
        return (nf, ns, facs) if anyd else None

    def _classify_distributed(self, fam_members):
        """Decide which candidate families run distributed over the lanes.  A family is kept
        distributed only if (a) every strided operand is element c of a distributed vector and every
        other operand is replicated, and (b) every reader consumes member c aligned: as operand of
        member c of another distributed family, or as term c of a reduction."""
        T = self.T
        P = self.P
        cand = {}
        for fid, m in enumerate(fam_members):
            if len(m) >= 2 * P:
                cand[fid] = m
        readers = {}
        for i in range(1, T.n + 1):
            if not T.used[i] or T.group[i] == 0:
                continue
            for j in set(self.operands(i)):
                readers.setdefault(j, []).append(i)
        changed = True
        while changed:
            changed = False
            # tentative distributed values of the surviving candidates
            for k in [k for k, v in self.dval.items() if v[0][0] == "F"]:
                del self.dval[k]
            member_of = {}
            for fid, m in cand.items():
                for c, i in enumerate(m):
                    self.dval[i] = (("F", fid), c)
                    member_of[i] = (fid, c)
            dfam_of = {("F", fid) for fid in cand}
            for fid, m in list(cand.items()):
                ok = True
                sig = None
                for c, i in enumerate(m):
                    leaves = self._leaf_ids(i)
                    row = []
                    for j in leaves:
                        if j in self.dval:
                            vec, el = self.dval[j]
                            if el != c or (vec[0] == "F" and vec not in dfam_of):
                                ok = False
                                break
                            row.append(("d", vec))
                        else:
                            row.append(("r", j if not T.isload[j - 1] else ("lit", float(T.pars[T.parStart[j - 1]]))))
                    if not ok:
                        break
                    if sig is None:
                        sig = row
                    elif row != sig:
                        ok = False
                        break
                    # inlined reductions inside a distributed member are not supported
                    if any(self._has_long_sum(i)):
                        ok = False
                        break
                    if self.out_needed[i]:
                        ok = False
                        break
                    for q in readers.get(i, []):
                        # reader chain: q may be inlined into its consumer; what matters is q itself
                        if q in member_of and member_of[q][1] == c and member_of[q][0] in cand:
                            continue
                        # inlined intermediate of a distributed member
                        top = self._consumer_root(q, readers)
                        if top in member_of and member_of[top][1] == c and member_of[top][0] in cand:
                            continue
                        if self._reduction_shape(q, dfam_of) is not None:
                            continue
                        ok = False
                        break
                    if not ok:
                        break
                if ok and not any(r[0] == "d" for r in (sig or [])):
                    ok = False        # nothing distributed to gain from
                if not ok:
                    del cand[fid]
                    changed = True
        # final values
        for k in [k for k, v in self.dval.items() if v[0][0] == "F"]:
            del self.dval[k]
        self._dcand = {}
        for fid, m in cand.items():
            for c, i in enumerate(m):
                self.dval[i] = (("F", fid), c)
            self.dvec_len[("F", fid)] = len(m)
            self._dcand[fid] = m
        dfam_of = {("F", fid) for fid in cand}
        for q in range(1, T.n + 1):
            if T.used[q] and T.group[q] > 0:
                r = self._reduction_shape(q, dfam_of)
                if r is not None:
                    self.dreduce[q] = r
        dm = set()
        for m in cand.values():
            dm.update(m)
        # distributed members are not SSA temporaries nor memory-resident
        for i in dm:
            self.local[i] = False
        return [m for fid, m in enumerate(fam_members) if fid not in cand]

    def _has_long_sum(self, i):
        for o in self.operands(i):
            if self.inline[o]:
                typ = int(self.T.types[o - 1])
                if typ in (I.I_sumprod, I.I_plus_sqr) and len(self.operands(o)) >= 2 * self.P:
                    yield True
                yield from self._has_long_sum(int(o))

    def _consumer_root(self, q, readers):
        """follow single-consumer inlining chains up to the instruction that is actually emitted"""
        while self.inline[q]:
            r = readers.get(q, [])
            if len(r) != 1:
                break
            q = r[0]
        return q

    def build_dfamilies(self):
        """materialise DFamily objects (after storage assignment)"""
        for fid, m in self._dcand.items():
            t = self.tree(m[0])
            leaves = []
            for lf in self.flat_leaves(t):
                if lf.kind == "dval":
                    leaves.append(("d", lf.idx[0]))
                else:
                    leaves.append(("r", lf))
            self.dfamilies[min(m)] = DFamily(fid, m, t, leaves)
            self.in_family[m] = True

    def _mine_families(self, fam_members):
        """same-shape instructions at the same dependency level of a group = mutually independent
        candidates, ordered by their operands so that storage indices have a chance to be affine"""
        T = self.T
        taken = set(self.dval)
        for m in fam_members:
            taken.update(m)
        out = []
        self._ilevel = np.zeros(T.n + 1, dtype=np.int64)
        for g in range(1, T.nGroups + 1):
            buckets = {}
            for i in T.groups[g - 1]:
                i = int(i)
                lv = 0
                for j in self._leaf_ids(i) if not self.inline[i] else self.operands(i):
                    if T.group[j] == g:
                        lv = max(lv, self._ilevel[j] + 1)
                self._ilevel[i] = lv
                if self.inline[i] or i in taken or not self.computed[i]:
                    continue
                buckets.setdefault((lv, self.shape_sig(i)), []).append(i)
            for key in sorted(buckets, key=lambda k: (k[0], buckets[k][0])):
                lst = buckets[key]
                if len(lst) >= MINE_MIN:
                    out.append(self._best_order(lst))
        return out

    def _best_order(self, lst):
        """order the (independent) members so that as many as possible fall into runs whose operand
        ids are arithmetic progressions: try each operand position as the primary sort key"""
        leaves = {i: tuple(self._leaf_ids(i)) for i in lst}
        nl = len(leaves[lst[0]])
        best, best_score = None, -1
        for p in range(max(1, nl)):
            order = sorted(lst, key=lambda i: (leaves[i][p:p + 1], leaves[i]))
            score, k0 = 0, 0
            while k0 < len(order):
                k1 = k0 + 1
                if k1 < len(order):
                    d = [b - a for a, b in zip(leaves[order[k0]], leaves[order[k1]])]
                    while k1 < len(order) and [b - a for a, b in zip(leaves[order[k1 - 1]], leaves[order[k1]])] == d:
                        k1 += 1
                if k1 - k0 >= MINE_MIN:
                    score += k1 - k0
                k0 = k1
            if score > best_score:
                best, best_score = order, score
        return best

    def _build_units(self, mined):
        """emission order of every group: units (family | distributed family | single instruction)
        sorted topologically, ties by lowest instruction id"""
        import heapq
        T = self.T
        unit_of = {}
        units = {}
        for key, f in self.families.items():
            units[("f", key)] = (f, f.members)
        for key, f in self.dfamilies.items():
            units[("d", key)] = (f, f.members)
        for uk, (_, mem) in units.items():
            for i in mem:
                unit_of[i] = uk
        self.units = []
        for g in range(1, T.nGroups + 1):
            glist = []
            for i in T.groups[g - 1]:
                i = int(i)
                if self.inline[i]:
                    continue
                uk = unit_of.get(i)
                if uk is None:
                    uk = ("1", i)
                    units[uk] = (i, [i])
                    unit_of[i] = uk
                if uk not in glist:
                    glist.append(uk)
            gset = set(glist)
            deps = {uk: set() for uk in glist}
            for uk in glist:
                for i in units[uk][1]:
                    for j in self._leaf_ids(i):
                        vk = unit_of.get(j)
                        if vk is not None and vk in gset and vk != uk:
                            deps[uk].add(vk)
            users = {uk: [] for uk in glist}
            for uk, ds in deps.items():
                for vk in ds:
                    users[vk].append(uk)
            ready = [(min(units[uk][1]), uk) for uk in glist if not deps[uk]]
            heapq.heapify(ready)
            order = []
            left = {uk: len(ds) for uk, ds in deps.items()}
            while ready:
                _, uk = heapq.heappop(ready)
                order.append(units[uk][0])
                for wk in users[uk]:
                    left[wk] -= 1
                    if left[wk] == 0:
                        heapq.heappush(ready, (min(units[wk][1]), wk))
            if len(order) != len(glist):
                raise AssertionError(f"group {g}: cyclic dependency between emission units")
            self.units.append(order)

    def _leaf_ids(self, i):
        out = []
        for o in self.operands(i):
            if self.inline[o]:
                out += self._leaf_ids(int(o))
            else:
                out.append(int(o))
        return out

    @staticmethod
    def flat_leaves(t, acc=None):
        if acc is None:
            acc = []
        for k in t.kids:
            if isinstance(k, Leaf):
                acc.append(k)
            else:
                Lowering.flat_leaves(k, acc)
        return acc

    def affine_trees(self, trees):
        """(base, delta) per flattened leaf if the trees share one structure and every leaf index is
        an arithmetic progression over the list, else None"""
        def sig(t):
            return (t.typ, t.par, tuple(sig(k) if isinstance(k, Node) else "L" for k in t.kids))
        s0 = sig(trees[0])
        L0 = self.flat_leaves(trees[0])
        if any(lf.kind == "dval" for lf in L0):
            return None
        n = len(trees)
        base = [lf.idx for lf in L0]
        delta = [0] * len(L0)
        if n > 1:
            L1 = self.flat_leaves(trees[1])
            if len(L1) != len(L0):
                return None
            delta = [(b.idx - a.idx) if a.kind in ("priv", "ext") and b.kind == a.kind else 0
                     for a, b in zip(L0, L1)]
        for c, t in enumerate(trees):
            if c and sig(t) != s0:
                return None
            Lc = self.flat_leaves(t)
            if len(Lc) != len(L0):
                return None
            for j, lf in enumerate(Lc):
                if lf.kind != L0[j].kind:
                    return None
                if lf.kind == "lit":
                    if lf.val != L0[j].val:
                        return None
                elif lf.kind == "tmp":
                    if lf.idx != L0[j].idx:
                        return None
                elif lf.idx != base[j] + delta[j] * c:
                    return None
        return base, delta

    def _family_verify(self, members):
        trees = [self.tree(i) for i in members]
        L0 = self.flat_leaves(trees[0])
        if any(lf.kind == "dval" for lf in L0):
            return None           # reads distributed values: members are emitted one by one
        n = len(members)
        base = [lf.idx for lf in L0]
        delta = [0] * len(L0)
        if n > 1:
            L1 = self.flat_leaves(trees[1])
            if len(L1) != len(L0):
                return None
            delta = [b.idx - a.idx for a, b in zip(L0, L1)]
        for c, t in enumerate(trees):
            Lc = self.flat_leaves(t)
            if len(Lc) != len(L0):
                return None
            for j, lf in enumerate(Lc):
                if lf.kind != L0[j].kind:
                    return None
                if lf.kind == "lit":
                    if lf.val != L0[j].val:
                        return None
                elif lf.kind == "tmp":
                    if lf.idx != L0[j].idx:
                        return None
                elif lf.idx != base[j] + delta[j] * c:
                    return None
        d = [int(self.pslot[i]) for i in members]
        if min(d) < 0:
            return None
        dd = d[1] - d[0] if n > 1 else 0
        if any(d[c] != d[0] + dd * c for c in range(n)):
            return None
        return Family(list(members), trees[0], base, delta, d[0], dd)

    # ---- V2: affine term families inside a long sum ---------------------------------------------------
    @staticmethod
    def term_family(node, leaf_base, leaf_delta, leaf_pos):
        """For a sumprod/plus_sqr ``node`` whose kids are all leaves: returns per-factor
        (kind, base, dk, dc) lists if terms form an arithmetic progression, else None.
        leaf_base/leaf_delta: per-leaf V1 base/delta (delta==0 outside families);
        leaf_pos: index of node's first leaf in the flattened leaf list."""
        if node.typ == I.I_sumprod:
            nf, ns = int(node.par[0]), int(node.par[1])
        elif node.typ == I.I_plus_sqr:
            nf, ns = 1, len(node.kids)
        else:
            return None
        if ns < V2_MIN or any(not isinstance(k, Leaf) for k in node.kids):
            return None
        out = []
        for p in range(nf):
            l0 = node.kids[p]
            b0 = leaf_base[leaf_pos + p]
            dc = leaf_delta[leaf_pos + p]
            dk = (leaf_base[leaf_pos + nf + p] - b0) if ns > 1 else 0
            for k in range(ns):
                lf = node.kids[k * nf + p]
                if lf.kind != l0.kind:
                    return None
                if lf.kind == "lit":
                    if lf.val != l0.val:
                        return None
                    continue
                if lf.kind == "tmp":
                    if lf.idx != l0.idx:
                        return None
                    continue
                if leaf_base[leaf_pos + k * nf + p] != b0 + dk * k or leaf_delta[leaf_pos + k * nf + p] != dc:
                    return None
            out.append((l0.kind, b0, dk, dc, l0))
        return nf, ns, out
