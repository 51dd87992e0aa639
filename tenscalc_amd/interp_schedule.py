"""Tape -> word stream of the table-driven device evaluator (csrc/tc_interp.hip.inc).

The reference prints every scalar instruction as a C statement (writeCinstructionsC,
/root/reference/lib/csparse/instructionsTableUTHash.c:950-1599).  For tapes of tens of thousands of
instructions the HIP back-end keeps them as data instead: per dependency group the instructions are
level-scheduled (an instruction's level = 1 + the highest level among its operands inside the group),
and same-shaped instructions of one level -- mutually independent by construction -- are packed into
bundles of 8 / 2 / 1 so that one bundle has all of its operand loads in flight at once.  Instruction
semantics are untouched: every instruction still evaluates its operands left to right exactly as the
reference prints them, so results do not depend on the packing.
"""
from __future__ import annotations

import numpy as np

from . import instructions as I

NB = 6          # max operands of an instruction inside an 8-wide record (TC_NB)
KB = 8          # record width
OP_COPY = 60    # pseudo-opcode: dest <- operand (copies / declareCopy entry points)
_LONG_OPS = {I.I_sum, I.I_sumprod, I.I_plus_sqr, I.I_plus_abs, I.I_min, I.I_min0, I.I_max, I.I_max0,
             I.I_max_abs, I.I_clp, I.I_minus_dot, I.I_minus_dot_div, I.I_plus_minus_dot,
             I.I_plus_minus_dot_div}
_DOT_SKIP = {I.I_minus_dot: 0, I.I_minus_dot_div: 1, I.I_plus_minus_dot: 1, I.I_plus_minus_dot_div: 2}


def _pad8(lst):
    return list(lst) + [0] * (-len(lst) % 8)


def _emit(words, op, n, aux, items):
    """items: list of (dest_row, [operand words row*2+neg]) with identical (op, n, aux).
    Record layouts: csrc/tc_interp.hip.inc."""
    if n > 8:
        if op not in _LONG_OPS:
            raise NotImplementedError(f"table-driven back-end: {n} operands for opcode {I.NAMES.get(op, op)}")
        skip = _DOT_SKIP.get(op, 0)
        for d, a in items:
            head = a[:skip] + [0, 0]
            body = _pad8(a[skip:])
            words += [op, n - skip, aux, d, head[0], head[1], 8 + len(body), 0] + body
        return
    k0 = 0
    while k0 < len(items):
        left = len(items) - k0
        if n <= NB and left >= 3:
            chunk = items[k0:k0 + KB]
            chunk = chunk + [chunk[-1]] * (KB - len(chunk))   # pad: recomputes the last one
            k0 += KB
            words += [op | (KB << 8), n, aux, 0, 0, 0, 16 + 8 * n, 0]
            words += [d for d, _ in chunk]
            for j in range(n):
                words += [a[j] for _, a in chunk]
        else:
            d, a = items[k0]
            k0 += 1
            words += [op | (1 << 8), n, aux, d, 0, 0, 16, 0] + _pad8(a)


def build_stream(T, row_of):
    """Returns (words int32 array, group ranges [(start,end)] per group, level statistics).

    row_of(j): state row (relative to the scratch base) holding the value of instruction j."""
    words = []
    ranges = []
    stats = []
    opl, ost = T.ops.tolist(), T.opStart.tolist()
    types = T.types
    grp = T.group
    for g, members in enumerate(T.groups, 1):
        start = len(words)
        lvl = {}
        levels = {}
        for i in members.tolist():
            l_ = 0
            for j in opl[ost[i - 1]:ost[i]]:
                if grp[j] == g:
                    l_ = max(l_, lvl[j] + 1)
            lvl[i] = l_
            levels.setdefault(l_, []).append(i)
        for l_ in sorted(levels):
            shapes = {}
            for i in levels[l_]:
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
                elif op >= I.I_componentwise or op in (I.I_set, I.I_load):
                    raise NotImplementedError(f"table-driven back-end: opcode {I.NAMES.get(op, op)}")
                else:
                    a = [row_of(j) * 2 for j in ops]
                d = row_of(i)
                if not a:               # I_2, I_zero: give the evaluator something to load
                    a = [d * 2]
                shapes.setdefault((op, len(a), aux), []).append((d, a))
            for (op, n, aux), items in sorted(shapes.items()):
                _emit(words, op, n, aux, items)
        ranges.append((start, len(words)))
        stats.append((len(members), len(levels)))
    return words, ranges, stats


def copy_stream(words, dst_rows, src_rows):
    """dest <- source for a copy entry point; 8 at a time when no destination is also a source"""
    start = len(words)
    items = [(int(d), [int(s) * 2]) for d, s in zip(dst_rows, src_rows)]
    if set(int(d) for d in dst_rows) & set(int(s) for s in src_rows):
        for d, a in items:              # keep the sequential element-by-element meaning
            words += [OP_COPY | (1 << 8), 1, 0, d, 0, 0, 16, 0] + _pad8(a)
    else:
        _emit(words, OP_COPY, 1, 0, items)
    return start, len(words)


def gather_table(words, rows):
    """plain rows, 8 per vector (tc_gather)"""
    start = len(words)
    words += _pad8([int(r) for r in rows])
    return start, len(rows)


def validate_stream(words, ranges, lo, hi):
    """Walk every record range exactly as csrc/tc_interp.hip.inc does and check that each row a lane
    would touch lies inside the tile (lo <= row < hi) and that the walk ends on the range end."""
    w = words
    for start, end in ranges:
        if start % 8 or end % 8:
            raise AssertionError("record range not 32-byte aligned")
        pc = start
        while pc < end:
            op, K = w[pc] & 255, w[pc] >> 8
            n, ln = w[pc + 1], w[pc + 6]
            if K == 8:
                if not (1 <= n <= NB) or ln != 16 + 8 * n:
                    raise AssertionError(f"bad 8-wide record at {pc}")
                rows = w[pc + 8:pc + 16] + [x >> 1 for x in w[pc + 16:pc + 16 + 8 * n]]
            elif K == 1:
                if not (1 <= n <= 8) or ln != 16:
                    raise AssertionError(f"bad single record at {pc}")
                rows = [w[pc + 3]] + [x >> 1 for x in w[pc + 8:pc + 8 + n]]
            elif K == 0:
                if n < 1 or ln != 8 + 8 * ((n + 7) // 8) or op not in _LONG_OPS:
                    raise AssertionError(f"bad long record at {pc}")
                rows = [w[pc + 3]] + [x >> 1 for x in w[pc + 8:pc + 8 + n]]
                skip = _DOT_SKIP.get(op, 0)
                rows += [x >> 1 for x in w[pc + 4:pc + 4 + skip]]
                if op in _DOT_SKIP and n % 2:
                    raise AssertionError(f"odd product operand count at {pc}")
            else:
                raise AssertionError(f"bad record width {K} at {pc}")
            if min(rows) < lo or max(rows) >= hi:
                raise AssertionError(f"row out of the tile in record at {pc}: {min(rows)}..{max(rows)} not in [{lo},{hi})")
            pc += ln
        if pc != end:
            raise AssertionError("record walk does not end on the range end")


def as_array(words):
    a = np.asarray(words, dtype=np.int64)
    if a.size and (a.max() >= 2 ** 31 or a.min() < -2 ** 31):
        raise OverflowError("stream word out of int32 range")
    return a.astype(np.int32)
