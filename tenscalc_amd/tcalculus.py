"""Symbolic tensor calculus front-end (host side, code-generation time only).

Mirrors the *interface* of the reference's ``@Tcalculus`` class
(/root/reference/lib/@Tcalculus/Tcalculus.m:64-202 hash-consed expression table,
``gradient.m:75-390`` derivative rules, ``Tcalculus.m:3049-3100`` hessian) with the
subset of operators that ``ipmPD_CStimesLambda.m`` and the four BASELINE problems
need.  It is a from-scratch design: expressions are immutable hash-consed nodes,
sizes are tuples (``()`` scalar, ``(n,)`` vector, ``(m,n)`` matrix, ...), and all
structural-sparsity work is deferred to :mod:`tenscalc_amd.csparse`.

MATLAB semantics kept on purpose so problem statements read like the reference's
examples: ``*`` is ``mtimes`` (matrix product, scalar product, vector'*vector inner
product for two 1-D operands), ``times(a,b)`` is element-wise, ``/`` is ``rdivide``,
``>=``/``<=``/``==`` build ``ispositive``/``iszero`` constraint nodes
(lib/private/parseConstraints.m:74-135).
"""
from __future__ import annotations

import re

import numpy as np

__all__ = [
    "Tcalculus", "Tvariable", "Tconstant", "Tzeros", "Tones", "Teye", "clear",
    "plus", "tprod", "mtimes", "times", "rdivide", "norm2", "norminf", "norm1",
    "compose", "componentwise", "tabs", "sqr", "sqrt", "cube", "exp", "log", "sin", "cos", "inv_",
    "heaviside", "relu", "tmin", "tmax", "min2", "max2", "clp", "cat", "subsref",
    "reshape", "repmat", "diag", "transpose", "ldl", "ldl_d", "ldl_l", "lu", "mldivide", "gradientVector",
    "full", "tsum", "gradient", "hessian", "substitute", "toT", "expand",
    "ispositive", "iszero", "vec",
]

_TABLE: dict = {}
_NODES: list = []
_GRAD_CACHE: dict = {}
_SUBST_CACHE: dict = {}


def clear():
    """Forget every expression (reference: ``Tcalculus.clear``, Tcalculus.m:95-120)."""
    _TABLE.clear()
    _NODES.clear()
    _GRAD_CACHE.clear()
    _SUBST_CACHE.clear()


def _prod(sz):
    p = 1
    for s in sz:
        p *= int(s)
    return p


class Tcalculus:
    __slots__ = ("op", "size", "args", "par", "idx", "value")

    def __init__(self, op, size, args, par, idx, value=None):
        self.op = op
        self.size = tuple(int(s) for s in size)
        self.args = tuple(args)
        self.par = par
        self.idx = idx
        self.value = value  # only for 'constant'

    # -- construction through the hash-consing table -------------------------
    @staticmethod
    def make(op, size, args=(), par=None, value=None):
        size = tuple(int(s) for s in size)
        key = (op, size, tuple(a.idx for a in args), par)
        n = _TABLE.get(key)
        if n is None:
            n = Tcalculus(op, size, args, par, len(_NODES), value)
            _NODES.append(n)
            _TABLE[key] = n
        return n

    # -- niceties ---------------------------------------------------------------
    @property
    def name(self):
        if self.op != "variable":
            raise AttributeError("only variables have names")
        return self.par

    @property
    def ndim(self):
        return len(self.size)

    def numel(self):
        return _prod(self.size)

    def __len__(self):
        if not self.size:
            return 1
        return self.size[0]

    def __repr__(self):
        if self.op == "variable":
            return f"T<{self.par}{list(self.size)}>"
        return f"T<{self.op}{list(self.size)}#{self.idx}>"

    def __hash__(self):
        return self.idx

    def __eq__(self, other):  # constraint a==b  (iszero)
        if isinstance(other, Tcalculus) and other is self:
            return True
        return iszero(plus([self, toT(other)], [1, -1]))

    def same(self, other):
        return self is other

    # -- operators (MATLAB semantics) ------------------------------------------
    def __add__(self, o):
        return plus([self, toT(o)], [1, 1])

    def __radd__(self, o):
        return plus([toT(o), self], [1, 1])

    def __sub__(self, o):
        return plus([self, toT(o)], [1, -1])

    def __rsub__(self, o):
        return plus([toT(o), self], [1, -1])

    def __neg__(self):
        return plus([self], [-1])

    def __pos__(self):
        return self

    def __mul__(self, o):
        return mtimes(self, toT(o))

    def __rmul__(self, o):
        return mtimes(toT(o), self)

    __matmul__ = __mul__
    __rmatmul__ = __rmul__

    def __truediv__(self, o):
        return rdivide(self, toT(o))

    def __rtruediv__(self, o):
        return rdivide(toT(o), self)

    def __ge__(self, o):
        return ispositive(plus([self, toT(o)], [1, -1]))

    def __le__(self, o):
        return ispositive(plus([toT(o), self], [1, -1]))

    __gt__ = __ge__
    __lt__ = __le__

    @property
    def T(self):
        return transpose(self)

    def __getitem__(self, key):
        if not isinstance(key, tuple):
            key = (key,)
        return subsref(self, key)


# ------------------------------------------------------------------------------
# leaves
# ------------------------------------------------------------------------------
def _size(sz):
    if sz is None:
        return ()
    if isinstance(sz, (int, np.integer)):
        return (int(sz),)
    return tuple(int(s) for s in sz)


def Tvariable(name, size=()):
    """``Tvariable name [size]`` (lib/Tvariable.m)."""
    return Tcalculus.make("variable", _size(size), (), str(name))


def Tzeros(size=()):
    return Tcalculus.make("zeros", _size(size))


def Tones(size=()):
    return Tcalculus.make("ones", _size(size))


def Teye(size):
    """Identity tensor; ``size`` = (n1..nk, n1..nk) (lib/Teye.m)."""
    size = _size(size)
    h = len(size) // 2
    if len(size) % 2 or size[:h] != size[h:]:
        raise ValueError(f"Teye: size {size} is not [n,n]")
    if not size:
        return Tones(())
    return Tcalculus.make("eye", size)


def Tconstant(value, size=None):
    """Numeric constant; all-zero / all-one values become zeros/ones nodes (lib/Tconstant.m)."""
    a = np.asarray(value, dtype=np.float64)
    if size is not None:
        size = _size(size)
        if a.size == 1 and _prod(size) != 1:
            a = np.full(size, float(a.reshape(-1)[0]))
        a = a.reshape(size, order="F")
    if a.ndim == 2 and a.shape[1] == 1 and size is None:
        a = a[:, 0]
    if a.size and not a.any():
        return Tzeros(a.shape)
    if a.size and (a == 1).all():
        return Tones(a.shape)
    if a.size == 0:
        return Tzeros(a.shape)
    a = np.array(a, dtype=np.float64, order="C").reshape(a.shape)
    return Tcalculus.make("constant", a.shape, (), (a.shape, a.tobytes()), value=a)


def toT(x):
    if isinstance(x, Tcalculus):
        return x
    return Tconstant(x)


def const_value(x):
    """Dense numpy value of a purely numeric node, or None."""
    if x.op == "constant":
        return x.value
    if x.op == "zeros":
        return np.zeros(x.size)
    if x.op == "ones":
        return np.ones(x.size)
    if x.op == "eye":
        h = len(x.size) // 2
        n = _prod(x.size[:h])
        return np.eye(n).reshape(x.size, order="F")
    return None


# ------------------------------------------------------------------------------
# plus
# ------------------------------------------------------------------------------
def expand(x, size):
    """Grow a scalar to ``size`` (reference grows scalars with reshape+subsref,
    Tcalculus.m:2327-2332); here a product with a ones tensor, which the
    scalarizer turns into pure aliasing."""
    size = _size(size)
    if x.size == size:
        return x
    if x.size != ():
        raise ValueError(f"cannot expand size {x.size} to {size}")
    if x.op == "zeros":
        return Tzeros(size)
    if x.op == "ones":
        return Tones(size)
    if x.op == "constant":
        return Tconstant(np.full(size, float(x.value)))
    return tprod(x, (), Tones(size), tuple(range(1, len(size) + 1)))


def plus(terms, signs):
    """n-ary signed sum, nested sums flattened and zero terms dropped
    (Tcalculus.m:2228-2310)."""
    terms = [toT(t) for t in terms]
    size = ()
    for t in terms:
        if t.size != ():
            if size != () and t.size != size:
                raise ValueError(f"plus: incompatible sizes {size} and {t.size}")
            size = t.size
    flat, fs = [], []
    for t, s in zip(terms, signs):
        t = expand(t, size)
        if t.op == "plus":
            for a, sa in zip(t.args, t.par):
                flat.append(a)
                fs.append(int(s) * int(sa))
        elif t.op == "zeros" or _prod(t.size) == 0:
            continue
        else:
            flat.append(t)
            fs.append(int(s))
    # merge numeric constants
    cacc = None
    keep, ks = [], []
    for t, s in zip(flat, fs):
        v = const_value(t)
        if v is not None:
            cacc = s * v if cacc is None else cacc + s * v
        else:
            keep.append(t)
            ks.append(s)
    if cacc is not None:
        c = Tconstant(cacc)
        if c.op != "zeros":
            keep.append(c)
            ks.append(1)
    if not keep:
        return Tzeros(size)
    if len(keep) == 1 and ks[0] == 1:
        return keep[0]
    return Tcalculus.make("plus", size, keep, tuple(ks))


# ------------------------------------------------------------------------------
# tprod
# ------------------------------------------------------------------------------
def tprod(*args):
    """``tprod(A,ia,B,ib,...)``: positive indices label output dimensions,
    negative indices are summed (doc/sphinx, Tcalculus tprod.m:1-75)."""
    if len(args) % 2:
        raise ValueError("tprod needs (tensor, indices) pairs")
    objs, inds = [], []
    for k in range(0, len(args), 2):
        o = toT(args[k])
        ix = args[k + 1]
        if isinstance(ix, (int, np.integer)):
            ix = (int(ix),)
        ix = tuple(int(i) for i in ix)
        if len(ix) != len(o.size):
            raise ValueError(f"tprod: operand of size {o.size} given {len(ix)} indices {ix}")
        objs.append(o)
        inds.append(ix)
    dims = {}
    for o, ix in zip(objs, inds):
        for i, s in zip(ix, o.size):
            if i == 0:
                raise ValueError("tprod: index 0 is not allowed")
            if dims.setdefault(i, s) != s:
                raise ValueError(f"tprod: index {i} has sizes {dims[i]} and {s}")
    npos = max([i for i in dims if i > 0], default=0)
    for i in range(1, npos + 1):
        if i not in dims:
            raise ValueError(f"tprod: output index {i} missing")
    nneg = max([-i for i in dims if i < 0], default=0)
    for i in range(1, nneg + 1):
        if -i not in dims:
            raise ValueError(f"tprod: summation index {-i} missing")
    osize = tuple(dims[i] for i in range(1, npos + 1))
    if any(o.op == "zeros" for o in objs) or any(s == 0 for s in dims.values()):
        return Tzeros(osize)
    # transposes are pure index relabeling
    for k in range(len(objs)):
        while objs[k].op == "transpose":
            objs[k] = objs[k].args[0]
            inds[k] = inds[k][::-1]
    # fold scalar numeric factors together
    c = 1.0
    keep_o, keep_i = [], []
    for o, ix in zip(objs, inds):
        if o.size == () and const_value(o) is not None:
            c *= float(const_value(o))
        else:
            keep_o.append(o)
            keep_i.append(ix)
    if c == 0.0:
        return Tzeros(osize)
    if c != 1.0 or not keep_o:
        keep_o.append(Tconstant(c) if c != 1.0 else Tones(()))
        keep_i.append(())
    objs, inds = keep_o, keep_i
    if len(objs) == 1 and inds[0] == tuple(range(1, len(inds[0]) + 1)):
        return objs[0]
    # canonical operand order: by node index (commutative)
    order = sorted(range(len(objs)), key=lambda k: (objs[k].idx, inds[k]))
    objs = [objs[k] for k in order]
    inds = [inds[k] for k in order]
    sums = tuple(dims[-i] for i in range(1, nneg + 1))
    return Tcalculus.make("tprod", osize, objs, (tuple(inds), sums))


def mtimes(a, b):
    """MATLAB ``*`` (Tcalculus.m:2437-2530)."""
    a, b = toT(a), toT(b)
    if a.size == ():
        return tprod(a, (), b, tuple(range(1, b.ndim + 1)))
    if b.size == ():
        return tprod(a, tuple(range(1, a.ndim + 1)), b, ())
    if a.ndim == 2 and b.ndim == 1:
        return tprod(a, (1, -1), b, (-1,))
    if a.ndim == 2 and b.ndim == 2:
        return tprod(a, (1, -1), b, (-1, 2))
    if a.ndim == 1 and b.ndim == 1:  # inner product
        return tprod(a, (-1,), b, (-1,))
    if a.ndim == 1 and b.ndim == 2:  # row-vector * matrix
        return tprod(a, (-1,), b, (-1, 1))
    raise ValueError(f"mtimes: sizes {a.size} x {b.size} not supported")


def times(a, b):
    """Element-wise product ``a.*b`` with scalar expansion."""
    a, b = toT(a), toT(b)
    if a.size == ():
        return tprod(a, (), b, tuple(range(1, b.ndim + 1)))
    if b.size == ():
        return tprod(a, tuple(range(1, a.ndim + 1)), b, ())
    if a.size != b.size:
        raise ValueError(f"times: sizes {a.size} and {b.size} differ")
    ix = tuple(range(1, a.ndim + 1))
    return tprod(a, ix, b, ix)


def rdivide(a, b):
    """``a./b`` for equal sizes or a scalar on either side (Tcalculus.m:2590-2656)."""
    a, b = toT(a), toT(b)
    if a.size != b.size and a.size != () and b.size != ():
        raise ValueError(f"rdivide: sizes {a.size} and {b.size} differ")
    size = a.size if a.size != () else b.size
    if a.op == "zeros":
        return Tzeros(size)
    if b.op == "ones" and b.size == ():
        return a
    return Tcalculus.make("rdivide", size, (a, b))


# ------------------------------------------------------------------------------
# reductions / element-wise functions
# ------------------------------------------------------------------------------
def norm2(x):
    """Sum of squares (``norm2``, Tcalculus.m:984-1004)."""
    x = toT(x)
    if x.op == "zeros":
        return Tzeros(())
    return Tcalculus.make("norm2", (), (x,))


def norminf(x):
    x = toT(x)
    if x.op == "zeros":
        return Tzeros(())
    return Tcalculus.make("norminf", (), (x,))


def norm1(x):
    x = toT(x)
    return Tcalculus.make("norm1", (), (x,))


# name -> (derivative name, preserves structural zeros?)
_FUNS = {
    "log": ("inv", False), "inv": ("minus_inv_sqr", False),
    "minus_inv_sqr": ("2_inv_cube", False), "2_inv_cube": (None, False),
    "exp": ("exp", False), "sin": ("cos", True), "cos": ("minus_sin", False),
    "minus_sin": ("minus_cos", True), "minus_cos": ("sin", False),
    "round": ("zero", True), "ceil": ("zero", True), "floor": ("zero", True),
    "abs": ("sign", True), "sign": ("zero", False),
    "sqrt": ("Dsqrt", True), "Dsqrt": ("DDsqrt", False), "DDsqrt": (None, False),
    "sqr": ("2times", True), "2times": ("2", True), "2": ("zero", False),
    "cube": ("3sqr", True), "3sqr": ("6times", True), "6times": (None, True),
    "atan": ("Datan", True), "Datan": ("DDatan", False), "DDatan": (None, True),
    "srelu": ("dsrelu", False), "dsrelu": ("ddsrelu", False), "ddsrelu": (None, False),
    "relu": ("heaviside", True), "heaviside": ("zero", True), "zero": ("zero", True),
}


def compose(x, fun):
    """Element-wise function application (reference ``compose``; the function table
    is the one in @csparse/computeScalarInstructions.m:366-410)."""
    x = toT(x)
    if fun not in _FUNS:
        raise ValueError(f"compose: unknown function {fun}")
    if fun == "zero":
        return Tzeros(x.size)
    if x.op == "zeros" and _FUNS[fun][1]:
        return Tzeros(x.size)
    return Tcalculus.make("compose", x.size, (x,), fun)


def componentwise(x, cfun):
    """Element-wise application of a scalar function given as C text with one ``%s`` hole for the operand, e.g.
    ``componentwise(x, "fabs(%s)")``: the reference's ``componentwise(obj,fun,cfun,...)`` (@Tcalculus/Tcalculus.m:
    2029-2031 uses it for abs), scalarized to I_componentwise (@csparse/sparsity_componentwise.m:35,
    instructionsTableUTHash.c:1333).  The result is structurally full; no derivative is defined."""
    x = toT(x)
    if cfun.count("%s") != 1:
        raise ValueError("componentwise: the C text needs exactly one %s hole")
    if not re.fullmatch(r"[A-Za-z0-9_+\-*/().,<>=?:%& |!]+", cfun):
        raise ValueError("componentwise: unsupported character in C text %r" % cfun)
    return Tcalculus.make("componentwise", x.size, (x,), cfun)


def tabs(x):
    return compose(x, "abs")


def sqr(x):
    return compose(x, "sqr")


def sqrt(x):
    return compose(x, "sqrt")


def cube(x):
    return compose(x, "cube")


def exp(x):
    return compose(x, "exp")


def log(x):
    return compose(x, "log")


def sin(x):
    return compose(x, "sin")


def cos(x):
    return compose(x, "cos")


def inv_(x):
    return compose(x, "inv")


def heaviside(x):
    return compose(x, "heaviside")


def relu(x):
    return compose(x, "relu")


def _reduce(op, x, dim):
    x = toT(x)
    if dim is None:
        dim = tuple(range(1, x.ndim + 1))
    if isinstance(dim, (int, np.integer)):
        dim = (int(dim),)
    dim = tuple(sorted(int(d) for d in dim))
    osize = tuple(s for k, s in enumerate(x.size, 1) if k not in dim)
    if not dim:
        return x
    return Tcalculus.make(op, osize, (x,), dim)


def tmin(x, dim=None):
    """``min(x,[],dim)`` (1-based dims; all dims when omitted)."""
    return _reduce("min", x, dim)


def tmax(x, dim=None):
    return _reduce("max", x, dim)


def min2(a, b):
    a, b = toT(a), toT(b)
    size = a.size if a.size != () else b.size
    return Tcalculus.make("min2", size, (expand(a, size), expand(b, size)))


def max2(a, b):
    a, b = toT(a), toT(b)
    size = a.size if a.size != () else b.size
    return Tcalculus.make("max2", size, (expand(a, size), expand(b, size)))


def clp(x, dx):
    """Largest step ``a>0`` keeping ``x+a*dx>=0`` (lib/clp.m)."""
    x, dx = toT(x), toT(dx)
    if x.size != dx.size:
        raise ValueError("clp: size mismatch")
    return Tcalculus.make("clp", (), (x, dx))


def tsum(x, dim=None):
    """``sum(x,dim)`` as a one-operand tprod (Tcalculus.m sum)."""
    x = toT(x)
    if dim is None:
        dim = tuple(range(1, x.ndim + 1))
    if isinstance(dim, (int, np.integer)):
        dim = (int(dim),)
    ix, p, q = [], 0, 0
    for k in range(1, x.ndim + 1):
        if k in dim:
            q += 1
            ix.append(-q)
        else:
            p += 1
            ix.append(p)
    return tprod(x, tuple(ix))


# ------------------------------------------------------------------------------
# structural (zero-flop) operators
# ------------------------------------------------------------------------------
def cat(dim, *xs):
    """Concatenate along 1-based ``dim`` (aliasing only, sparsity_cat.m)."""
    xs2 = []
    for x in xs:
        x = toT(x)
        if x.ndim == 0:
            x = reshape(x, (1,))
        if x.ndim >= dim and x.size[dim - 1] == 0:
            continue
        xs2.append(x)
    xs = xs2
    if not xs:
        return Tzeros((0,))
    nd = max(x.ndim for x in xs)
    if nd < dim:
        raise ValueError("cat: dim exceeds tensor order")
    base = None
    total = 0
    for x in xs:
        if x.ndim != nd:
            raise ValueError("cat: mismatched orders")
        other = x.size[: dim - 1] + x.size[dim:]
        if base is None:
            base = other
        elif base != other:
            raise ValueError(f"cat: mismatched sizes {base} vs {other}")
        total += x.size[dim - 1]
    osize = base[: dim - 1] + (total,) + base[dim - 1:]
    if len(xs) == 1:
        return xs[0]
    if all(x.op == "zeros" for x in xs):
        return Tzeros(osize)
    return Tcalculus.make("cat", osize, xs, int(dim))


def vec(x):
    x = toT(x)
    return reshape(x, (_prod(x.size),))


def subsref(x, key):
    """``x(i1,i2,...)`` with 0-based Python indexing: ints drop the dimension,
    slices and integer sequences keep it."""
    x = toT(x)
    key = tuple(key)
    if len(key) > x.ndim:
        raise IndexError("too many indices")
    key = key + (slice(None),) * (x.ndim - len(key))
    subs, osize, drop = [], [], []
    for k, (kk, n) in enumerate(zip(key, x.size)):
        if isinstance(kk, slice):
            ii = tuple(range(*kk.indices(n)))
            subs.append(ii)
            osize.append(len(ii))
        elif isinstance(kk, (int, np.integer)):
            i = int(kk)
            if i < 0:
                i += n
            if not 0 <= i < n:
                raise IndexError("index out of range")
            subs.append((i,))
            osize.append(1)
            drop.append(k)
        else:
            ii = tuple(int(i) + (n if int(i) < 0 else 0) for i in kk)
            if any(not 0 <= i < n for i in ii):
                raise IndexError("index out of range")
            subs.append(ii)
            osize.append(len(ii))
    full_id = all(s == tuple(range(n)) for s, n in zip(subs, x.size))
    if full_id:
        y = x
    elif x.op == "zeros":
        y = Tzeros(osize)
    elif x.op == "ones":
        y = Tones(osize)
    else:
        y = Tcalculus.make("subsref", osize, (x,), tuple(subs))
    if drop:
        y = reshape(y, tuple(s for k, s in enumerate(osize) if k not in drop))
    return y


def reshape(x, size):
    x = toT(x)
    size = _size(size)
    if _prod(size) != _prod(x.size):
        raise ValueError(f"reshape: {x.size} -> {size}")
    if size == x.size:
        return x
    if x.op == "zeros":
        return Tzeros(size)
    if x.op == "ones":
        return Tones(size)
    if x.op == "reshape":
        return reshape(x.args[0], size)
    if x.op == "constant":
        return Tconstant(x.value.reshape(size, order="F"))
    return Tcalculus.make("reshape", size, (x,))


def repmat(x, reps):
    """``repmat``: implemented through index replication (subsref)."""
    x = toT(x)
    reps = _size(reps)
    if x.ndim == 0:
        x = reshape(x, (1,) * len(reps))
    if len(reps) != x.ndim:
        raise ValueError("repmat: reps must have one entry per dimension")
    key = tuple(tuple(list(range(n)) * r) for n, r in zip(x.size, reps))
    return subsref(x, key)


def diag(x):
    """vector -> diagonal matrix, square matrix -> diagonal vector (sparsity_diag.m)."""
    x = toT(x)
    if x.ndim == 1:
        if x.op == "zeros":
            return Tzeros((x.size[0], x.size[0]))
        return Tcalculus.make("diag", (x.size[0], x.size[0]), (x,))
    if x.ndim == 2 and x.size[0] == x.size[1]:
        if x.op == "zeros":
            return Tzeros((x.size[0],))
        return Tcalculus.make("diag", (x.size[0],), (x,))
    raise ValueError("diag: needs a vector or a square matrix")


def transpose(x):
    x = toT(x)
    if x.ndim <= 1:
        return x
    if x.ndim != 2:
        raise ValueError("transpose: only for matrices")
    if x.op == "transpose":
        return x.args[0]
    if x.op == "zeros":
        return Tzeros(x.size[::-1])
    if x.op == "ones":
        return Tones(x.size[::-1])
    if x.op == "eye":
        return x
    return Tcalculus.make("transpose", x.size[::-1], (x,))


def full(x):
    x = toT(x)
    if x.op == "full":
        return x
    return Tcalculus.make("full", x.size, (x,))


# ------------------------------------------------------------------------------
# factorizations
# ------------------------------------------------------------------------------
def ldl(A, tag="", stages=None):
    """Symbolic LDL' with a fixed (codegen-time) elimination order
    (Tcalculus.m:1394-1486, @csparse/sparsity_ldl.m).  ``stages`` (one int per row) constrains
    the fill-reducing order: all rows of stage s are eliminated before any row of stage s+1
    (used to keep the tiny -eps*I pivots of equality duals behind the primal block)."""
    A = toT(A)
    if A.ndim != 2 or A.size[0] != A.size[1]:
        raise ValueError("ldl: needs a square matrix")
    if stages is not None:
        stages = tuple(int(s) for s in stages)
        if len(stages) != A.size[0]:
            raise ValueError("ldl: one stage per row expected")
    return Tcalculus.make("ldl", A.size, (A,), (str(tag), stages))


def ldl_d(f):
    if f.op != "ldl":
        raise ValueError("ldl_d: needs an ldl() factorization")
    return Tcalculus.make("ldl_d", (f.size[0],), (f,))


def ldl_l(f):
    if f.op != "ldl":
        raise ValueError("ldl_l: needs an ldl() factorization")
    return Tcalculus.make("ldl_l", f.size, (f,))


def lu(A, tag=""):
    """Symbolic LU with row and column orders fixed at code-generation time from typical values
    (Tcalculus.m ``lu``, @csparse/sparsity_lu.m): used by the equilibrium solvers, whose Newton matrix is not
    symmetric (lib/ipmPDeqlat_CS.m:653)."""
    A = toT(A)
    if A.ndim != 2 or A.size[0] != A.size[1]:
        raise ValueError("LU decomposition only implemented for square matrices")
    return Tcalculus.make("lu", A.size, (A,), (str(tag),))


def mldivide(f, b):
    """``ldl(A)\\b`` (split into l1/d/u1 solves by csparse, csparse.m:906-957) or ``lu(A)\\b`` (l1/u solves,
    csparse.m:880-902)."""
    b = toT(b)
    if f.op not in ("ldl", "lu"):
        f = ldl(f)
    if b.ndim not in (1, 2) or b.size[0] != f.size[0]:
        raise ValueError("mldivide: size mismatch")
    if b.op == "zeros":
        return Tzeros(b.size)
    return Tcalculus.make("mldivide", b.size, (f, b))


# ------------------------------------------------------------------------------
# constraints
# ------------------------------------------------------------------------------
def ispositive(x):
    return Tcalculus.make("ispositive", toT(x).size, (toT(x),))


def iszero(x):
    return Tcalculus.make("iszero", toT(x).size, (toT(x),))


# ------------------------------------------------------------------------------
# derivatives  (gradient.m:75-390)
# ------------------------------------------------------------------------------
def gradient(f, var):
    """d f / d var, size = size(f) + size(var)."""
    f = toT(f)
    if var.op != "variable":
        raise ValueError("gradient: only with respect to variables")
    key = (f.idx, var.idx)
    g = _GRAD_CACHE.get(key)
    if g is None:
        g = _gradient(f, var)
        if g.size != f.size + var.size:
            raise AssertionError(f"gradient size {g.size} != {f.size + var.size} for {f.op}")
        _GRAD_CACHE[key] = g
    return g


def _gradient(f, var):
    gsize = f.size + var.size
    nv = var.ndim
    nf = f.ndim
    op = f.op
    if op in ("zeros", "ones", "eye", "constant"):
        return Tzeros(gsize)
    if op == "variable":
        return Teye(gsize) if f is var else Tzeros(gsize)
    gs = [gradient(a, var) for a in f.args] if op not in ("mldivide", "ldl", "ldl_d", "ldl_l", "lu") else None
    vix = tuple(range(1, nv + 1))
    if op == "plus":
        return plus(gs, f.par)
    if op == "tprod":
        inds, _ = f.par
        out = []
        for i, (gi, ix) in enumerate(zip(gs, inds)):
            if gi.op == "zeros":
                continue
            a = [gi, ix + tuple(nf + k for k in vix)]
            for j, (o, jx) in enumerate(zip(f.args, inds)):
                if j != i:
                    a += [o, jx]
            out.append(tprod(*a))
        return plus(out, [1] * len(out)) if out else Tzeros(gsize)
    if op == "transpose":
        return gradient(tprod(f.args[0], (2, 1)), var)
    if op == "norm2":
        x = f.args[0]
        neg = tuple(-k for k in range(1, x.ndim + 1))
        return tprod(Tconstant(2.0), (), gs[0], neg + vix, x, neg)
    if op == "componentwise":
        raise NotImplementedError("componentwise(%r) has no derivative" % f.par)
    if op == "compose":
        d = _FUNS[f.par][0]
        if d is None:
            raise NotImplementedError(f"derivative of {f.par} not available")
        x = f.args[0]
        if gs[0].op == "zeros" or d == "zero":
            return Tzeros(gsize)
        p = tuple(range(1, x.ndim + 1))
        return tprod(compose(x, d), p, gs[0], p + tuple(x.ndim + k for k in vix))
    if op == "rdivide":
        a, b = f.args
        ga, gb = gs
        if a.size == () and b.size == ():
            I = ()
        else:
            I = tuple(range(1, nf + 1))
        K = tuple(nf + k for k in vix)
        out, sg = [], []
        if ga.op != "zeros":
            if a.size == () and b.size != ():
                out.append(tprod(ga, K, compose(b, "inv"), I))
            elif b.size == ():
                out.append(rdivide(ga, b))
            else:
                out.append(tprod(ga, I + K, compose(b, "inv"), I))
            sg.append(1)
        if gb.op != "zeros":
            if b.size == ():
                out.append(rdivide(tprod(a, tuple(range(1, a.ndim + 1)), gb, K), times(b, b)))
            elif a.size == ():
                out.append(tprod(gb, I + K, rdivide(a, times(b, b)), I))
            else:
                out.append(tprod(rdivide(a, times(b, b)), I, gb, I + K))
            sg.append(-1)
        return plus(out, sg) if out else Tzeros(gsize)
    if op == "subsref":
        return subsref(gs[0], f.par + (slice(None),) * nv)
    if op == "reshape":
        return reshape(gs[0], gsize)
    if op == "full":
        return full(gs[0])
    if op == "cat":
        return cat(f.par, *gs)
    if op == "diag":
        x = f.args[0]
        n = f.size[0]
        if x.ndim == 1:  # vector -> matrix: D[i,j,v] = eye3[i,j,k] g[k,v]
            e3 = _eye3(n)
            return tprod(e3, (1, 2, -1), gs[0], (-1,) + tuple(2 + k for k in vix))
        e3 = _eye3(n)     # matrix -> vector: d[i,v] = sum_jk eye3[j,k,i] g[j,k,v]
        return tprod(e3, (-1, -2, 1), gs[0], (-1, -2) + tuple(1 + k for k in vix))
    if op in ("min", "max", "min2", "max2", "clp", "norminf", "norm1", "ldl", "ldl_d", "ldl_l", "lu"):
        raise NotImplementedError(f"gradient of {op} not implemented (not smooth)")
    if op == "mldivide":
        F, b = f.args
        A = F.args[0]
        gA = gradient(A, var)
        gb = gradient(b, var)
        # d(A\b) = A\(db - dA*(A\b))  (gradient.m:211-217)
        K = tuple(range(nf + 1, nf + nv + 1))
        rhs = gb
        if gA.op != "zeros":
            if b.ndim == 1:
                rhs = plus([gb, tprod(gA, (1, -1) + K, f, (-1,))], [1, -1])
            else:
                rhs = plus([gb, tprod(gA, (1, -1) + K, f, (-1, 2))], [1, -1])
        if rhs.op == "zeros":
            return Tzeros(gsize)
        cols = _prod(rhs.size[1:])
        sol = mldivide(F, reshape(rhs, (rhs.size[0], cols)))
        return reshape(sol, gsize)
    raise NotImplementedError(f"gradient of '{op}' not implemented")


def _eye3(n):
    a = np.zeros((n, n, n))
    i = np.arange(n)
    a[i, i, i] = 1.0
    return Tconstant(a)


def gradientVector(obj, variables):
    """lib/gradientVector.m: the gradients of a vector (or scalar) expression with respect to several vector variables,
    side by side -- [n, sum of the variables' lengths] (empty variables are skipped)."""
    obj = toT(obj)
    gs = [gradient(obj, v) for v in variables if _prod(v.size) > 0]
    if not gs:
        return Tzeros(tuple(obj.size) + (0,))
    return gs[0] if len(gs) == 1 else cat(obj.ndim + 1, *gs)


def hessian(f, var, var2=None):
    """``[H,g]=hessian(f,u)`` (Tcalculus.m:3049-3100)."""
    g = gradient(f, var)
    H = gradient(g, var if var2 is None else var2)
    return H, g


# ------------------------------------------------------------------------------
# substitution
# ------------------------------------------------------------------------------
def substitute(f, var, expr):
    """Replace variable ``var`` by ``expr`` throughout ``f`` (Tcalculus.m substitute).
    ``f`` may be a list; ``var``/``expr`` may be parallel lists."""
    if isinstance(f, (list, tuple)):
        return type(f)(substitute(x, var, expr) for x in f)
    if isinstance(var, (list, tuple)):
        pairs = tuple((v.idx, toT(e)) for v, e in zip(var, expr))
    else:
        pairs = ((var.idx, toT(expr)),)
    mp = {i: e for i, e in pairs}
    ctx = tuple((i, e.idx) for i, e in pairs)
    return _subst(toT(f), mp, ctx)


def _subst(f, mp, ctx):
    key = (f.idx, ctx)
    r = _SUBST_CACHE.get(key)
    if r is not None:
        return r
    if f.op == "variable":
        r = mp.get(f.idx, f)
        if r.size != f.size:
            raise ValueError(f"substitute: size mismatch for {f}")
    elif not f.args:
        r = f
    else:
        na = [_subst(a, mp, ctx) for a in f.args]
        if all(x is y for x, y in zip(na, f.args)):
            r = f
        else:
            r = _rebuild(f, na)
    _SUBST_CACHE[key] = r
    return r


def _rebuild(f, a):
    op = f.op
    if op == "plus":
        return plus(a, f.par)
    if op == "tprod":
        inds, _ = f.par
        args = []
        for o, ix in zip(a, inds):
            args += [o, ix]
        return tprod(*args)
    if op == "rdivide":
        return rdivide(*a)
    if op == "norm2":
        return norm2(a[0])
    if op == "norminf":
        return norminf(a[0])
    if op == "norm1":
        return norm1(a[0])
    if op == "compose":
        return compose(a[0], f.par)
    if op == "componentwise":
        return componentwise(a[0], f.par)
    if op in ("min", "max"):
        return _reduce(op, a[0], f.par)
    if op == "min2":
        return min2(*a)
    if op == "max2":
        return max2(*a)
    if op == "clp":
        return clp(*a)
    if op == "cat":
        return cat(f.par, *a)
    if op == "subsref":
        return subsref(a[0], f.par)
    if op == "reshape":
        return reshape(a[0], f.size)
    if op == "diag":
        return diag(a[0])
    if op == "transpose":
        return transpose(a[0])
    if op == "full":
        return full(a[0])
    if op == "ldl":
        return ldl(a[0], f.par[0], f.par[1])
    if op == "lu":
        return lu(a[0], f.par[0])
    if op == "ldl_d":
        return ldl_d(a[0])
    if op == "ldl_l":
        return ldl_l(a[0])
    if op == "mldivide":
        return mldivide(*a)
    if op == "ispositive":
        return ispositive(a[0])
    if op == "iszero":
        return iszero(a[0])
    raise NotImplementedError(f"substitute through '{op}'")


def depends_on(f, var, _seen=None):
    """True when ``f`` references variable ``var``."""
    if _seen is None:
        _seen = {}
    r = _seen.get(f.idx)
    if r is None:
        r = f is var or any(depends_on(a, var, _seen) for a in f.args)
        _seen[f.idx] = r
    return r


def variables_of(f, _acc=None, _seen=None):
    if _acc is None:
        _acc, _seen = [], set()
    if f.idx in _seen:
        return _acc
    _seen.add(f.idx)
    if f.op == "variable":
        _acc.append(f)
    for a in f.args:
        variables_of(a, _acc, _seen)
    return _acc
