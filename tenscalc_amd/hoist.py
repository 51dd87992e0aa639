"""Opt-in expression rewrite: contract parameter-only factors first.

TensCalc evaluates a ``tprod`` chain in the association order the derivative rules produce
(@Tcalculus/gradient.m:180-184: d norm2(y) = 2*tprod(dy,[-1,1],y,[-1])), so the gradient of a
least-squares cost ``norm2(A*x-b)`` is recomputed every Newton iteration as ``A'*(A*x-b)``:
two passes over ``A``.  Its dependency groups already hoist what depends on parameters only
(``A'*A`` for the Hessian is evaluated once per ``setP_A``, dependencyGroups.m:111-116); this pass
makes more of the tape parameter-only by two algebraic identities on the expression graph:

  R1  tprod(P.., plus(t1,..,tn)) -> plus(tprod(P..,t1),..,tprod(P..,tn))   (linearity)
  R2  tprod(P.., tprod(Q.., x..)) -> tprod(tprod(P..,Q..), x..)           (re-association)
  R3  norm2(t1+..+tn) -> sum_s <ts,ts> + 2 sum_{s<s'} <ts,ts'>            (then R1/R2 on every inner product)

where P.., Q.. depend only on *slow* variables (the class parameters) and x.. on anything else.
A rewrite is kept only if it lowers the per-iteration multiply-add count (dense estimate).  For
sls the gradient becomes ``(2A'A) x - 2A'b`` with both coefficients evaluated once per solve, and the
cost ``norm2(A*x-b)`` becomes ``x'(A'A)x - 2(A'b)'x + b'b``: after the coefficients have been formed
nothing reads A or b again (one pass over the inputs per solve).  R3 trades a sum of squares for a
difference of larger numbers: the relative rounding error of the cost grows by about
(|x'A'Ax|+2|b'Ax|+|b'b|)/|cost|, a few hundred for the benchmark data, i.e. ~1e-13 instead of 1e-16 --
far inside the tolerance these classes are tested at, but it is a change of arithmetic like the others.

The rewritten tape is a different sequence of fp64 operations: results agree with the reference
tape to rounding (not bit for bit), so classes built with ``hoist=True`` are tested at the stated
tolerance of the reassociating lowerings (tests/test_hoist.py, tests/test_gpu_parity.py), never
against the bit-exact golden vectors.  Default everywhere: off.
"""
from __future__ import annotations

from . import tcalculus as tc

__all__ = ["Hoister"]


class Hoister:
    def __init__(self, slow_vars):
        self.slow_vars = {v.idx for v in slow_vars}
        self._slow = {}
        self._param = {}
        self._memo = {}
        self.log = []           # (kind, before idx, after idx, cost before, cost after)

    # ---- classification ---------------------------------------------------------------
    def is_slow(self, f):
        """True when ``f`` depends on slow variables (and constants) only."""
        r = self._slow.get(f.idx)
        if r is None:
            if f.op == "variable":
                r = f.idx in self.slow_vars
            else:
                r = all(self.is_slow(a) for a in f.args)
            self._slow[f.idx] = r
        return r

    def has_param(self, f):
        """True when ``f`` references at least one slow variable (i.e. is not a pure constant)."""
        r = self._param.get(f.idx)
        if r is None:
            if f.op == "variable":
                r = f.idx in self.slow_vars
            else:
                r = any(self.has_param(a) for a in f.args)
            self._param[f.idx] = r
        return r

    def fast_cost(self, f, _seen=None):
        """Dense multiply-add estimate of the nodes of ``f`` that must be re-evaluated whenever a
        fast variable changes."""
        top = _seen is None
        if top:
            _seen = set()
        if f.idx in _seen or self.is_slow(f):
            return 0
        _seen.add(f.idx)
        c = sum(self.fast_cost(a, _seen) for a in f.args)
        if f.op == "tprod":
            inds, sums = f.par
            dims = {}
            for o, ix in zip(f.args, inds):
                for i, s in zip(ix, o.size):
                    dims[i] = s
            nterms = 1
            for s in dims.values():
                nterms *= s
            c += nterms * max(1, len(f.args) - 1)
        elif f.op not in ("variable", "reshape", "subsref", "cat", "transpose", "full", "diag"):
            c += max(1, tc._prod(f.size)) * max(1, len(f.args))
        return c

    # ---- rewriting -----------------------------------------------------------------------
    def rewrite(self, f):
        f = tc.toT(f)
        r = self._memo.get(f.idx)
        if r is not None:
            return r
        if not f.args or f.op in ("ldl", "ldl_d", "ldl_l", "mldivide"):
            r = f
        elif self.is_slow(f):
            r = self._split_scalars(f) if f.op == "tprod" else f
            if r is f and f.op != "tprod":
                na = [self.rewrite(a) for a in f.args]
                r = f if all(x is y for x, y in zip(na, f.args)) else tc._rebuild(f, na)
        else:
            na = [self.rewrite(a) for a in f.args]
            r = f if all(x is y for x, y in zip(na, f.args)) else tc._rebuild(f, na)
            if r.op == "tprod" and not self.is_slow(r):
                r = self._improve(r)
            elif r.op == "norm2" and not self.is_slow(r):
                r = self._expand_norm2(r)
        self._memo[f.idx] = r
        self._memo[r.idx] = r
        return r

    def _expand_norm2(self, f):
        """R3, kept only if the expanded form needs fewer per-evaluation multiply-adds"""
        u = f.args[0]
        if u.op != "plus" or not any(self.has_param(t) for t in u.args):
            return f
        idx = tuple(-(k + 1) for k in range(u.ndim))
        pieces, signs = [], []
        terms = list(zip(u.args, u.par))
        for a in range(len(terms)):
            ta, sa = terms[a]
            pieces.append(self.rewrite(tc.tprod(ta, idx, ta, idx)))
            signs.append(1)
            for b in range(a + 1, len(terms)):
                tb, sb = terms[b]
                pieces.append(self.rewrite(tc.tprod(tc.Tconstant(2.0), (), ta, idx, tb, idx)))
                signs.append(int(sa) * int(sb))
        cand = tc.plus(pieces, signs)
        cb, ca = self.fast_cost(f), self.fast_cost(cand)
        if ca < cb:
            self.log.append(("expand norm2", f.idx, cand.idx, cb, ca))
            return cand
        return f

    def _split_scalars(self, f):
        """slow c*tprod(A,B,..) -> c*(tprod(A,B,..)): the same canonical form _reassociate gives its
        slow block, so e.g. the Hessian 2*A'A and the hoisted gradient coefficient share one A'A"""
        inds, _ = f.par
        scal = [k for k, o in enumerate(f.args) if o.size == ()]
        big = [k for k, o in enumerate(f.args) if o.size != ()]
        if not scal or len(big) < 2 or not self.has_param(f):
            return f
        margs = []
        for k in big:
            margs += [f.args[k], inds[k]]
        M = tc.tprod(*margs)
        sargs = []
        for k in scal:
            sargs += [f.args[k], ()]
        return tc.tprod(*(sargs + [M, tuple(range(1, M.ndim + 1))]))

    def _improve(self, f):
        best, cbest = f, self.fast_cost(f)
        for kind, cand in (("distribute", self._distribute(f)), ("reassociate", self._reassociate(f))):
            if cand is None or cand is f:
                continue
            c = self.fast_cost(cand)
            if c < cbest:
                self.log.append((kind, f.idx, cand.idx, cbest, c))
                best, cbest = cand, c
        return best

    def _distribute(self, f):
        """R1 on the first fast ``plus`` factor, provided another factor carries a parameter."""
        inds, _ = f.par
        if not any(self.is_slow(o) and self.has_param(o) and o.size != () for o in f.args):
            return None
        for j, o in enumerate(f.args):
            if o.op == "plus" and not self.is_slow(o):
                terms = []
                for t in o.args:
                    a = []
                    for k, (p, ix) in enumerate(zip(f.args, inds)):
                        a += [t if k == j else p, ix]
                    terms.append(self.rewrite(tc.tprod(*a)))
                return self._factor_out_scalar(terms, o.par)
        return None

    @staticmethod
    def _factor_out_scalar(terms, signs):
        """c*a - c*b -> c*(a - b) for a numeric scalar c common to all terms: otherwise a parameter-only
        term like 2*(A'b) is stored next to the A'b other expressions share"""
        def split(t):
            if t.op != "tprod":
                return None
            inds, _ = t.par
            for k, o in enumerate(t.args):
                if o.size == () and tc.const_value(o) is not None:
                    rest = []
                    for j, (p, ix) in enumerate(zip(t.args, inds)):
                        if j != k:
                            rest += [p, ix]
                    return float(tc.const_value(o)), (tc.tprod(*rest) if rest else tc.Tones(()))
            return None
        parts = [split(t) for t in terms]
        if len(terms) < 2 or any(p is None for p in parts) or len({p[0] for p in parts}) != 1 or parts[0][0] == 1.0:
            return tc.plus(terms, signs)
        inner = tc.plus([p[1] for p in parts], signs)
        return tc.tprod(tc.Tconstant(parts[0][0]), (), inner, tuple(range(1, inner.ndim + 1)))

    def _reassociate(self, f):
        """R2: flatten fast tprod factors into this tprod, then contract all slow factors first."""
        inds, sums = f.par
        objs, labs = [], []
        nneg = len(sums)
        flattened = False
        for o, ix in zip(f.args, inds):
            if o.op == "tprod" and not self.is_slow(o):
                iinds, isums = o.par
                for p, jx in zip(o.args, iinds):
                    objs.append(p)
                    labs.append(tuple(ix[i - 1] if i > 0 else -(nneg - i) for i in jx))
                nneg += len(isums)
                flattened = True
            else:
                objs.append(o)
                labs.append(tuple(ix))
        slow = [k for k in range(len(objs)) if self.is_slow(objs[k])]
        fast = [k for k in range(len(objs)) if not self.is_slow(objs[k])]
        big = [k for k in slow if objs[k].size != ()]
        if not fast or len(big) < 2 or not any(self.has_param(objs[k]) for k in big):
            return None
        fast_labels = {i for k in fast for i in labs[k]}
        slow_labels = []
        for k in slow:
            for i in labs[k]:
                if i not in slow_labels:
                    slow_labels.append(i)
        outer_l = [i for i in slow_labels if i > 0 or i in fast_labels]   # outputs of the slow block M
        inner_l = [i for i in slow_labels if i < 0 and i not in fast_labels]
        mmap = {i: n + 1 for n, i in enumerate(sorted(outer_l, key=lambda i: (i < 0, abs(i))))}
        mmap.update({i: -(n + 1) for n, i in enumerate(inner_l)})
        # the contraction of the non-scalar slow factors first: sum_k A_ki*A_kj*2 would cost one more
        # multiplication per term
        margs = []
        for k in big:
            margs += [objs[k], tuple(mmap[i] for i in labs[k])]
        M = tc.tprod(*margs)
        scal = [k for k in slow if objs[k].size == ()]     # stay in the outer product: 2*(A'A) would be a second
        #                                                    stored copy of A'A next to the one other uses share
        m_out = tuple(sorted(outer_l, key=lambda i: mmap[i]))
        # renumber the summation labels that survive in the outer product
        keep_neg = []
        for i in list(m_out) + [i for k in fast for i in labs[k]]:
            if i < 0 and i not in keep_neg:
                keep_neg.append(i)
        omap = {i: -(n + 1) for n, i in enumerate(keep_neg)}
        rel = lambda t: tuple(omap.get(i, i) for i in t)                 # noqa: E731
        oargs = [M, rel(m_out)] if M.size != () or tc.const_value(M) is None else [M, ()]
        for k in scal:
            oargs += [objs[k], ()]
        for k in fast:
            oargs += [objs[k], rel(labs[k])]
        return tc.tprod(*oargs)
