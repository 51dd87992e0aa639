"""Tape -> word stream of the table-driven device evaluator (csrc/tc_interp.hip.inc).

The reference prints every scalar instruction as a C statement (writeCinstructionsC,
/root/reference/lib/csparse/instructionsTableUTHash.c:950-1599).  For tapes of tens of thousands of
instructions the HIP back-end keeps them as data instead: per dependency group the instructions are
level-scheduled (an instruction's level = 1 + the highest level among its operands inside the group),
and same-shaped instructions of one level -- mutually independent by construction -- are packed into
records of 16 / 8 / 1 so that one record has all of its operand loads in flight at once.  Instruction
semantics are untouched: every instruction still evaluates its operands left to right exactly as the
reference prints them, so results do not depend on the packing.
"""
from __future__ import annotations

import os

import numpy as np

from . import instructions as I

NB = 8          # max operands of an instruction inside an 8-wide record (TC_NB)
KB = 8          # record width
KB2, NB2 = 16, 3  # wider records for instructions with few operands
OP_COPY = 60    # pseudo-opcode: dest <- operand (copies / declareCopy entry points)
_LONG_OPS = {I.I_sum, I.I_sumprod, I.I_plus_sqr, I.I_plus_abs, I.I_min, I.I_min0, I.I_max, I.I_max0,
             I.I_max_abs, I.I_clp, I.I_minus_dot, I.I_minus_dot_div, I.I_plus_minus_dot,
             I.I_plus_minus_dot_div}
_DOT_SKIP = {I.I_minus_dot: 0, I.I_minus_dot_div: 1, I.I_plus_minus_dot: 1, I.I_plus_minus_dot_div: 2}


LONG_CHUNK = 32   # TC_CH: operands of a long instruction are fetched this many at a time


def _pad8(lst, m=8):
    return list(lst) + [0] * (-len(lst) % m)


OP_BARRIER = 2 << 8   # header word 0 of a barrier record (K code 2)
K_LONG4 = 4           # K code: 4 long instructions of one shape side by side
L4_CHUNK = 8          # TC_C4: their operands are fetched this many at a time, per instruction


def _knob(name, default):
    """tuning knobs of the record packing (hashed into the class build key by hip_emitter.build_class)"""
    return float(os.environ.get(name, default))


def _records(op, n, aux, items, W=1):
    """items: list of (dest_row, [operand words row*2+neg]) with identical (op, n, aux) and no
    dependence on one another -> list of (cost, words) records.  Layouts: csrc/tc_interp.hip.inc."""
    out = []
    if n > 8:
        if op not in _LONG_OPS:
            raise NotImplementedError(f"table-driven back-end: {n} operands for opcode {I.NAMES.get(op, op)}")
        skip = _DOT_SKIP.get(op, 0)
        # four of a shape side by side (4 x 8 loads in flight per round instead of one instruction's 32, most of which
        # are padding when n is 9..16), as long as every wave still gets some
        l4 = _knob("TC_INTERP_L4_MIN", 1)
        if l4 > 0 and len(items) >= max(4, l4 * W):
            k0 = 0
            while len(items) - k0 >= 2:
                chunk = items[k0:k0 + 4]
                chunk = chunk + [chunk[-1]] * (4 - len(chunk))
                k0 += 4
                m = n - skip
                nch = (m + L4_CHUNK - 1) // L4_CHUNK
                heads = [a[:skip] + [0, 0] for _, a in chunk]
                w = [op | (K_LONG4 << 8), m, aux, 0, 0, 0, 24 + 32 * nch, 0]
                w += [d for d, _ in chunk] + [h[0] for h in heads] + [h[1] for h in heads] + [0, 0, 0, 0]
                bodies = [_pad8(a[skip:], L4_CHUNK) for _, a in chunk]
                for c in range(nch):
                    for b in bodies:
                        w += b[L4_CHUNK * c:L4_CHUNK * (c + 1)]
                out.append((4 * n + 24 * nch, w))
            items = items[k0:]
        for d, a in items:
            head = a[:skip] + [0, 0]
            body = _pad8(a[skip:], LONG_CHUNK)      # padded with row 0: loads need no guards
            out.append((n + 24, [op, n - skip, aux, d, head[0], head[1], 8 + len(body), 0] + body))
        return out
    # wide records only while every wave of the block still gets a few of them
    wm = _knob("TC_INTERP_WIDE_MIN", 3)
    if n <= NB2 and len(items) >= max(12, 4 * wm * W):
        K = KB2
    elif n <= NB and len(items) >= max(3, wm * W):
        K = KB
    else:
        K = 1
    k0 = 0
    while k0 < len(items):
        left = len(items) - k0
        if K > 1 and left >= 3:
            kk = K if left > KB or K == KB else KB
            chunk = items[k0:k0 + kk]
            chunk = chunk + [chunk[-1]] * (kk - len(chunk))   # pad: recomputes the last one
            k0 += kk
            w = [op | (kk << 8), n, aux, 0, 0, 0, 8 + kk + kk * n, 0]
            w += [d for d, _ in chunk]
            for j in range(n):
                w += [a[j] for _, a in chunk]
            out.append((kk * n + 24, w))
        else:
            d, a = items[k0]
            k0 += 1
            out.append((n + 24, [op | (1 << 8), n, aux, d, 0, 0, 16, 0] + _pad8(a)))
    return out


def _spread(level_records, W, streams):
    """one level's records over the W waves of a block (longest first onto the least loaded wave),
    then a barrier record in every wave's stream"""
    if W == 1:
        for _, w in level_records:
            streams[0] += w
        return
    load = [0] * W
    for cost, w in sorted(level_records, key=lambda r: -r[0]):
        k = load.index(min(load))
        load[k] += cost
        streams[k] += w
    for k in range(W):
        streams[k] += [OP_BARRIER, 0, 0, 0, 0, 0, 8, 0]


# ---- long reductions in parallel parts (round 6) ---------------------------------------------------------------------------
# A long record is ONE wavefront fetching 32 operands, waiting a memory round trip, folding them, fetching the next 32: a
# 1 992-operand sum of products is 63 round trips while the other wavefronts of the tile wait at the level's barrier (l1l2: ten
# such records per Newton iteration, each alone in its level).  They are cut into parts that ordinary records of the same level
# evaluate side by side, and one more record combines the parts in order -- with the reference's result, bit for bit:
#   I_sumprod  the reference prints a long sum of products as statements of <= MAX_TERMS_PERLINE = 100 factors,
#              m[r] = line1; m[r] += line2; ... (instructionsTableUTHash.c:173,1003-1012): the parts ARE those lines, the
#              combining record is the ordered sum line1 + line2 + ...
#   I_min / I_max / I_max_abs   `if (acc > x) acc = x` from the first operand (:1236-1313): a NaN later in the list is skipped,
#              a NaN in front sticks, of equal values (+0 / -0) the first stays.  Part 0 is the chain over the first operands;
#              the other parts start from a constant that never wins (+inf, -inf, 0 for absolute values), so a NaN at their
#              head is skipped as the reference skips it; the combining record is the same chain over the parts, in order
#   I_clp      from DBL_MAX (:1315-1331): every part is the chain over its own (x, dx) pairs from DBL_MAX, combined by I_min
SPLIT_LONG_OPS = {I.I_sumprod, I.I_min, I.I_max, I.I_max_abs, I.I_clp}


def split_long(op, aux, a, rows, W=8):
    """operand words `a` of one long instruction -> ([(op, aux, operand words, dest row)] parts, (op, aux, operand words) of
    the combining instruction), or None when the instruction is left alone.  rows: the _ExtraRows of the class."""
    n = len(a)
    if n < int(_knob("TC_INTERP_SPLIT_LONG", 256)) or op not in SPLIT_LONG_OPS:
        return None
    if op == I.I_sumprod:
        if aux < 1 or n % aux:
            return None
        nterm, bounds, start, nS = n // aux, [], 0, 0
        for t in range(nterm - 1):              # TcSumprod::step: the line is closed behind a term once 100 factors are in it
            nS += aux
            if nS >= 100:
                bounds.append((start, (t + 1) * aux))
                start, nS = (t + 1) * aux, 0
        bounds.append((start, n))
        if len(bounds) < 2:
            return None
        parts = [(I.I_sumprod, aux, a[s_:e_], rows.new_row()) for s_, e_ in bounds]
        return parts, (I.I_sum, 0, [d * 2 for _, _, _, d in parts])
    unit = 2 if op == I.I_clp else 1
    P = min(W, n // (64 * unit))
    if P < 2:
        return None
    lead = {I.I_min: float("inf"), I.I_max: float("-inf"), I.I_max_abs: 0.0}.get(op)
    nl = 0 if lead is None else 1
    m = -(-(n + nl * (P - 1)) // (P * LONG_CHUNK)) * LONG_CHUNK     # words per part, the leading constant included: whole fetch chunks
    parts, s_ = [], 0
    while s_ < n:
        head = [rows.const(lead) * 2] if (s_ and nl) else []
        e_ = s_ + m - len(head)
        parts.append((op, aux, head + a[s_:e_], rows.new_row()))
        s_ = e_
    return parts, (I.I_min if op == I.I_clp else op, 0, [d * 2 for _, _, _, d in parts])


class _ExtraRows:
    """rows above the tape's constants: constants the split records lead with, and the rows of their parts (initialised like
    constants by the kernel prologue, then written before they are read in every evaluation of their group)"""

    def __init__(self, base):
        self.base, self.vals, self._c = base, [], {}

    def const(self, v):
        if v not in self._c:
            self._c[v] = self.base + len(self.vals)
            self.vals.append(v)
        return self._c[v]

    def new_row(self):
        self.vals.append(0.0)
        return self.base + len(self.vals) - 1


NA_PAIR, NB_PAIR = 6, 3   # operand counts of the producer / the consumer inside a pair record (tc_pair<NA, NB>)
K_PAIR = 3                # K code of a pair record
# opcodes tc_pair_vals evaluates (csrc/tc_interp.hip.inc): the shapes that make up nearly all single-reader producers
PAIR_PRODUCERS = {I.I_sum, I.I_sumprod, I.I_div, I.I_sqr, I.I_2times}
PAIR_CONSUMERS = {I.I_sum, I.I_sumprod, I.I_div, I.I_plus_minus_dot, I.I_abs, I.I_heaviside}


def _pair_records(key, items):
    """items: (dest row of the consumer, producer operand words, consumer operand words -- the word at position p carries
    only the sign bit) of one pair shape -> records of 8 pairs (the last one padded by repeating its last pair)"""
    opB, nB, auxB, opA, nA, auxA, p = key
    out = []
    for k0 in range(0, len(items), KB):
        chunk = items[k0:k0 + KB]
        chunk = chunk + [chunk[-1]] * (KB - len(chunk))
        w = [opB | (K_PAIR << 8), nB, auxB, opA | (nA << 8) | (auxA << 16), p, 0, 16 + KB * (nA + nB), 0]
        w += [d for d, _, _ in chunk]
        for j in range(nA):
            w += [a[j] for _, a, _ in chunk]
        for j in range(nB):
            w += [b[j] for _, _, b in chunk]
        out.append((KB * (nA + nB - 1) + 24, w))
    return out


def build_stream(T, row_of, W=1, fusable=None, extra=None):
    """Returns (words, ranges, level statistics): ranges[g] = [(start, end)] * W, the record range
    each of the W cooperating waves walks for dependency group g+1.

    row_of(j): state row (relative to the scratch base) holding the value of instruction j.
    fusable[j] (optional): instruction j has exactly one reader, in its own group, and no entry point reads it.  Such a
    producer is evaluated INSIDE its consumer's record (a pair record: 8 producer/consumer pairs of one shape) and never
    stored: one store and one load of a 512-byte row less per pair.  The arithmetic of both instructions is untouched.
    extra (an _ExtraRows, W > 1): long reductions are evaluated in parallel parts + one combining record (split_long)."""
    words = []
    ranges = []
    stats = []
    opl, ost = T.ops.tolist(), T.opStart.tolist()
    types = T.types
    grp = T.group
    cache = {}

    def enc(i):
        """(opcode, aux, operand words, dest row) of instruction i as the evaluator sees it"""
        if i in cache:
            return cache[i]
        op = int(types[i - 1])
        ops = opl[ost[i - 1]:ost[i]]
        par = T.pars[T.parStart[i - 1]:T.parStart[i]]
        aux = 0
        if op == I.I_sum:
            a = [row_of(j) * 2 + (0 if p > 0 else 1) for j, p in zip(ops, par)]
            if len(a) != len(par):
                raise AssertionError("I_sum sign count")
        elif op == I.I_sumprod:
            aux = int(par[0])
            if aux * int(par[1]) != len(ops):
                raise AssertionError("I_sumprod operand count")
            a = [row_of(j) * 2 for j in ops]
        elif op == I.I_componentwise:
            eq = I.componentwise_opcode(par)
            if eq is None:
                raise NotImplementedError("table-driven back-ends run I_componentwise only for plain calls of "
                                          + str(sorted(I.COMPONENTWISE_CALLS)) + "; got " + repr(I.componentwise_text(par)))
            op = eq
            a = [int(row_of(j)) * 2 for j in ops]
        elif op > I.I_componentwise or op in (I.I_set, I.I_load):
            raise NotImplementedError(f"table-driven back-end: opcode {I.NAMES.get(op, op)}")
        else:
            a = [row_of(j) * 2 for j in ops]
        d = row_of(i)
        if not a:               # I_2, I_zero: give the evaluator something to load
            a = [d * 2]
        cache[i] = (op, aux, a, d)
        return cache[i]

    for g, members in enumerate(T.groups, 1):
        lvl = {}
        levels = {}
        for i in members.tolist():
            l_ = 0
            for j in opl[ost[i - 1]:ost[i]]:
                if grp[j] == g:
                    l_ = max(l_, lvl[j] + 1)
            lvl[i] = l_
            levels.setdefault(l_, []).append(i)
        # ---- producer/consumer pairs -----------------------------------------------------------------------------------
        host = {}       # consumer -> (producer, operand position)
        if fusable is not None:
            cand = {}   # (level of the consumer, pair shape) -> [consumer]
            for i in members.tolist():
                ops = opl[ost[i - 1]:ost[i]]
                if not ops or len(ops) > NB_PAIR or enc(i)[0] not in PAIR_CONSUMERS:
                    continue
                for p, j in enumerate(ops):
                    nj = ost[j] - ost[j - 1] if j >= 1 else 0
                    if not (fusable[j] and grp[j] == g and j not in host and 1 <= nj <= NA_PAIR):
                        continue
                    opB, auxB, aB, _ = enc(i)
                    opA, auxA, aA, _ = enc(j)
                    if opA not in PAIR_PRODUCERS:
                        continue
                    host[i] = (j, p)
                    cand.setdefault((lvl[i], (opB, len(aB), auxB, opA, len(aA), auxA, p)), []).append(i)
                    break
            for (_, _), lst in cand.items():
                # a shape needs a few full records per wave; the ragged tail of a shape stays unfused
                keep = len(lst) if len(lst) >= 3 * W else 0
                if keep % KB and keep % KB < 3:
                    keep -= keep % KB
                for i in lst[keep:]:
                    del host[i]
        inside = {j for j, _ in host.values()}
        streams = [[] for _ in range(W)]
        npair = 0
        nsplit = 0
        for l_ in sorted(levels):
            shapes = {}
            pshapes = {}
            post = {}               # combining records of this level's split reductions: a level of their own
            for i in levels[l_]:
                if i in inside:
                    continue
                op, aux, a, d = enc(i)
                sp = split_long(op, aux, a, extra, W) if (extra is not None and W > 1 and i not in host) else None
                if sp is not None:
                    for pop, paux, pa, pd in sp[0]:
                        shapes.setdefault((pop, len(pa), paux), []).append((pd, pa))
                    fop, faux, fa = sp[1]
                    post.setdefault((fop, len(fa), faux), []).append((d, fa))
                    nsplit += 1
                    continue
                if i in host:
                    j, p = host[i]
                    opA, auxA, aA, _ = enc(j)
                    b = list(a)
                    b[p] &= 1                # the producer's value never has a row: keep the sign bit only
                    pshapes.setdefault((op, len(a), aux, opA, len(aA), auxA, p), []).append((d, aA, b))
                    npair += 1
                else:
                    shapes.setdefault((op, len(a), aux), []).append((d, a))
            recs = []
            for (op, n, aux), items in sorted(shapes.items()):
                recs += _records(op, n, aux, items, W)
            for key, items in sorted(pshapes.items()):
                recs += _pair_records(key, items)
            if recs:
                _spread(recs, W, streams)
            if post:
                recs = []
                for (op, n, aux), items in sorted(post.items()):
                    recs += _records(op, n, aux, items, W)
                _spread(recs, W, streams)
        rg = []
        for k in range(W):
            rg.append((len(words), len(words) + len(streams[k])))
            words += streams[k]
        ranges.append(rg)
        stats.append((len(members), len(levels), npair, nsplit))
    return words, ranges, stats


def copy_stream(words, dst_rows, src_rows, W=1):
    """dest <- source for a copy entry point -> [(start, end)] * W.  8 at a time spread over the
    waves when no destination is also a source, else element by element on wave 0."""
    items = [(int(d), [int(s) * 2]) for d, s in zip(dst_rows, src_rows)]
    streams = [[] for _ in range(W)]
    if set(int(d) for d in dst_rows) & set(int(s) for s in src_rows):
        recs = []
        for d, a in items:              # keep the sequential element-by-element meaning
            streams[0] += [OP_COPY | (1 << 8), 1, 0, d, 0, 0, 16, 0] + _pad8(a)
        _spread([], W, streams)
    else:
        _spread(_records(OP_COPY, 1, 0, items, W), W, streams)
    rg = []
    for k in range(W):
        rg.append((len(words), len(words) + len(streams[k])))
        words += streams[k]
    return rg


def gather_table(words, rows):
    """plain rows, 8 per vector (tc_gather)"""
    start = len(words)
    words += _pad8([int(r) for r in rows])
    return start, len(rows)


def _record_rows(w, pc):
    """(rows a lane touches, length) of the record at pc; raises on a malformed record"""
    op, K = w[pc] & 255, w[pc] >> 8
    n, ln = w[pc + 1], w[pc + 6]
    if K in (KB, KB2):
        if not (1 <= n <= (NB if K == KB else NB2)) or ln != 8 + K + K * n:
            raise AssertionError(f"bad {K}-wide record at {pc}")
        return w[pc + 8:pc + 8 + K] + [x >> 1 for x in w[pc + 8 + K:pc + 8 + K + K * n]], ln
    if K == 1:
        if not (1 <= n <= 8) or ln != 16:
            raise AssertionError(f"bad single record at {pc}")
        return [w[pc + 3]] + [x >> 1 for x in w[pc + 8:pc + 8 + n]], ln
    if K == 0:
        if n < 1 or ln != 8 + LONG_CHUNK * ((n + LONG_CHUNK - 1) // LONG_CHUNK) or op not in _LONG_OPS:
            raise AssertionError(f"bad long record at {pc}")
        if op in _DOT_SKIP and n % 2:
            raise AssertionError(f"odd product operand count at {pc}")
        skip = _DOT_SKIP.get(op, 0)
        return ([w[pc + 3]] + [x >> 1 for x in w[pc + 8:pc + 8 + n]] + [x >> 1 for x in w[pc + 4:pc + 4 + skip]]), ln
    if K == K_LONG4:
        nch = (n + L4_CHUNK - 1) // L4_CHUNK
        if n < 1 or ln != 24 + 32 * nch or op not in _LONG_OPS:
            raise AssertionError(f"bad 4-long record at {pc}")
        if op in _DOT_SKIP and n % 2:
            raise AssertionError(f"odd product operand count at {pc}")
        skip = _DOT_SKIP.get(op, 0)
        rows = w[pc + 8:pc + 12] + [x >> 1 for x in w[pc + 12:pc + 12 + 4 * skip]]
        for c in range(nch):
            for q in range(4):
                v = w[pc + 24 + 32 * c + 8 * q:pc + 24 + 32 * c + 8 * (q + 1)]
                rows += [x >> 1 for x in v]          # padding words name row 0, which the device does load
        return rows, ln
    if K == K_PAIR:
        nA, p = (w[pc + 3] >> 8) & 255, w[pc + 4]
        if not (1 <= n <= NB_PAIR and 1 <= nA <= NA_PAIR and 0 <= p < n) or ln != 16 + KB * (nA + n):
            raise AssertionError(f"bad pair record at {pc}")
        rows = w[pc + 8:pc + 16] + [x >> 1 for x in w[pc + 16:pc + 16 + KB * nA]]
        for j in range(n):
            v = w[pc + 16 + KB * (nA + j):pc + 16 + KB * (nA + j + 1)]
            if j == p:
                if any(x >> 1 for x in v):
                    raise AssertionError(f"pair record at {pc}: the producer's position names a row")
            else:
                rows += [x >> 1 for x in v]
        return rows, ln
    raise AssertionError(f"bad record width {K} at {pc}")


def validate_stream(words, ranges, lo, hi):
    """Walk every record range exactly as csrc/tc_interp.hip.inc does and check that each row a lane
    would touch lies inside the tile (lo <= row < hi), that the walk ends on the range end, and that
    the waves sharing a range meet the same number of barriers.  ranges: [[(start, end)] * W]."""
    w = words
    for rg in ranges:
        nbar = set()
        for start, end in rg:
            if start % 8 or end % 8:
                raise AssertionError("record range not 32-byte aligned")
            pc, bars = start, 0
            while pc < end:
                if w[pc] == OP_BARRIER:
                    if w[pc + 6] != 8:
                        raise AssertionError(f"bad barrier record at {pc}")
                    bars += 1
                    pc += 8
                    continue
                rows, ln = _record_rows(w, pc)
                if min(rows) < lo or max(rows) >= hi:
                    raise AssertionError(f"row out of the tile in record at {pc}: {min(rows)}..{max(rows)} "
                                         f"not in [{lo},{hi})")
                pc += ln
            if pc != end:
                raise AssertionError("record walk does not end on the range end")
            nbar.add(bars)
        if len(nbar) > 1:
            raise AssertionError("waves of one range do not meet the same number of barriers")


ROW_BYTES = 512       # one row of a tile: 64 instances x 8 bytes


def device_words(words, ranges, lo):
    """The stream as the device walks it.  The builder, the validator and the CPU model of the walk speak in ROWS (dest
    words are rows, operand words row*2+neg); the device addresses the tile through a buffer descriptor whose base is the
    tile's first row, so every dest / operand word of every record becomes that row's BYTE OFFSET (row - lo) * 512, with
    the sign bit of an I_sum operand still in bit 0: one scalar AND and one buffer_load per operand instead of three scalar
    shifts, a 64-bit vector add and a global_load.  Words outside the record ranges (the output gather table) stay rows."""
    w = list(words)
    done = set()

    def off(row):
        b = (row - lo) * ROW_BYTES
        if not 0 <= b < 2 ** 31:            # every converted word, header words included: an offset that wrapped would
            raise OverflowError("byte offset of a row outside [0, 2 GB)")     # silently address another row of the tile
        return b

    def dst(k):
        w[k] = off(words[k])

    def opd(k):
        w[k] = off(words[k] >> 1) | (words[k] & 1)

    for rg in ranges:
        for start, end in rg:
            pc = start
            while pc < end:
                op, K, n, ln = words[pc] & 255, words[pc] >> 8, words[pc + 1], words[pc + 6]
                if pc in done:
                    raise AssertionError("record in two ranges")
                done.add(pc)
                if K in (KB, KB2):
                    for k in range(pc + 8, pc + 8 + K):
                        dst(k)
                    for k in range(pc + 8 + K, pc + 8 + K + K * n):
                        opd(k)
                elif K == 1:
                    dst(pc + 3)
                    for k in range(pc + 8, pc + 8 + n):
                        opd(k)
                elif K == 0:
                    dst(pc + 3)
                    for k in range(pc + 4, pc + 4 + _DOT_SKIP.get(op, 0)):
                        opd(k)
                    for k in range(pc + 8, pc + ln):          # the padding names row 0: the device does load it
                        opd(k)
                elif K == K_LONG4:
                    for k in range(pc + 8, pc + 12):
                        dst(k)
                    for k in range(pc + 12, pc + 20):         # heads (row 0 where the opcode has none)
                        opd(k)
                    for k in range(pc + 24, pc + ln):
                        opd(k)
                elif K == K_PAIR:
                    nA, p = (words[pc + 3] >> 8) & 255, words[pc + 4]
                    for k in range(pc + 8, pc + 16):
                        dst(k)
                    for k in range(pc + 16, pc + 16 + KB * nA):
                        opd(k)
                    for j in range(n):
                        if j != p:                             # position p carries sign bits only
                            for k in range(pc + 16 + KB * (nA + j), pc + 16 + KB * (nA + j + 1)):
                                opd(k)
                elif words[pc] != OP_BARRIER:
                    raise AssertionError(f"bad record width {K} at {pc}")
                pc += ln
    return w


def shape_masks(words, ranges):
    """Which record shapes and opcodes a class's stream contains: {"16_2": opcode bit mask, "8_3": ..., "1_7": ..., "long": ...,
    "long4": ..., "pairs": bit (nA-1)*3+(nB-1)}.  hip_emitter prints them as TC_OPS_* macros and csrc/tc_interp.hip.inc
    instantiates only those shapes, each with only those opcode arms (the dispatcher of a class is a third of the generic
    one: less to compile, and the arms a wavefront does run sit closer together in the instruction cache)."""
    m = {}
    for rg in ranges:
        for start, end in rg:
            pc = start
            while pc < end:
                op, K, n = words[pc] & 255, words[pc] >> 8, words[pc + 1]
                if op >= 64:
                    raise AssertionError("opcode outside the 64-bit masks")
                if K in (KB, KB2, 1):
                    key = "%d_%d" % (K, n)
                    m[key] = m.get(key, 0) | (1 << op)
                elif K == 0:
                    m["long"] = m.get("long", 0) | (1 << op)
                elif K == K_LONG4:
                    m["long4"] = m.get("long4", 0) | (1 << op)
                elif K == K_PAIR:
                    nA = (words[pc + 3] >> 8) & 255
                    m["pairs"] = m.get("pairs", 0) | (1 << ((nA - 1) * NB_PAIR + (n - 1)))
                pc += words[pc + 6]
    return m


def constant_rows(T, L):
    """constants read by instructions get rows of their own above the computed values"""
    consts, const_row = [], {}
    for i in range(1, T.n + 1):
        if T.isload[i - 1] and T.used[i]:
            const_row[i] = L.nPriv + len(consts)
            consts.append(float(T.pars[T.parStart[i - 1]]))
    return consts, const_row


def class_stream(T, L, classname, W=1):
    """Everything the device walk needs for one class: group records, copy records, the output
    gather table, and the row map (L = Lowering(T, spill_all=True))."""
    consts, const_row = constant_rows(T, L)

    def row_of(j):
        lf = L.leaf(int(j))
        if lf.kind == "lit":
            return const_row[int(j)]
        if lf.kind == "ext":
            return int(lf.idx) - L.nExt       # inputs sit just below the scratch rows
        if lf.kind == "priv":
            return int(lf.idx)
        raise AssertionError(lf.kind)
    fusable = L.single_local if os.environ.get("TC_INTERP_PAIRS", "1") != "0" else None
    extra = _ExtraRows(L.nPriv + len(consts)) if W > 1 else None
    words, granges, gstats = build_stream(T, row_of, W, fusable, extra)
    if extra is not None:
        consts = consts + extra.vals          # (a new list: the rows above the tape's constants)
    copy_rng = {}
    for e in T.copies:
        if any(L.external[d] for d in e["dst"]):
            raise NotImplementedError("copy into an external set")
        keep = [(row_of(d), row_of(s_)) for d, s_ in zip(e["dst"], e["src"]) if L.pslot[d] >= 0]
        copy_rng[e["name"]] = copy_stream(words, [d for d, _ in keep], [s_ for _, s_ in keep], W)
    go = T.entry(f"{classname}_getOutputs")[2]
    # every row the device walk will touch must be inside the tile: checked here, on the host
    lo, hi = -L.nExt, L.nPriv + len(consts)
    validate_stream(words, granges + list(copy_rng.values()), lo, hi)
    out_rows = [row_of(i) for o in go["outputs"] for i in o["ins"]]
    if out_rows and (min(out_rows) < lo or max(out_rows) >= hi):
        raise AssertionError("output row outside the tile")
    out_rng = gather_table(words, out_rows)
    dev_words = device_words(words, granges + list(copy_rng.values()), lo)
    shapes = shape_masks(words, granges + list(copy_rng.values()))
    # range table the device indexes by (range id, wave): groups first, then the copies
    rid = {}
    table = []
    for g, rg in enumerate(granges, 1):
        rid[g] = len(table) // (2 * W)
        table += [x for se in rg for x in se]
    for name, rg in copy_rng.items():
        rid[name] = len(table) // (2 * W)
        table += [x for se in rg for x in se]
    return dict(words=words, dev_words=dev_words, shapes=shapes, granges=granges, gstats=gstats, copy_rng=copy_rng, out_rng=out_rng,
                consts=consts, const_row=const_row, row_of=row_of, lo=lo, hi=hi, W=W, rid=rid, table=table)


def as_array(words):
    a = np.asarray(words, dtype=np.int64)
    if a.size and (a.max() >= 2 ** 31 or a.min() < -2 ** 31):
        raise OverflowError("stream word out of int32 range")
    return a.astype(np.int32)
