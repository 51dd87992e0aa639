"""Computation-graph files (.cg/.cgc/.cgio/.cgs) for scalarized tapes: the wire format a MATLAB TensCalc
installation already has (``saveScalarized(csparseObject, basename)``), so it can hand ANY problem's tape to the HIP
back-end without new MATLAB code paths.

File layout as documented in /root/reference/doc/computationgraphs.tex:84-220 and written by
/root/reference/lib/@csparse/saveScalarized.m:103-321: four files, each starting with a one-byte magic (32 / 64 =
little-endian int32 / int64 index words; saveScalarized writes int32, :16) followed by variable-length records.

* ``.cgc``  constants: (type, byte length) in index words, then the payload (always int64 / double): 10001
  double_full_tensor [nd, size..., values], 20001 int64l_full_vector [values], 20002 int64l_full_tensor
  [nd, size..., values], 30001 string  (type ids: lib/CGregistration.m:14-24).
* ``.cg``   one record per function node: (type, nIn, nOut, inputs..., outputs...), 0-based variable ids = instruction
  index - 1; type 0 = constant node (constant index, output variable).  Scalar function types are the registry's
  (CGregistration.m:67-130): 20001 Ssum, 20002 Ssumprod, 20003 Sinv, ... 20032 SDDatan -- NOT opcode numbers, and
  not the stale CGregistration.json (which lists 20002 for every scalar function).  I_sum / I_sumprod take their
  parameter vector as FIRST input: a variable produced by a constant node (saveScalarized.m:26-49,176-183).
  30001 scalars2ftensor (subscripts constant variable + scalar variables -> one full-tensor variable) and 30002
  ftensor2scalars (one full-tensor variable -> scalar variables) wrap the gets and sets (:203-247).
* ``.cgio`` gateway functions: (1 set | 2 get | 3 copy, n, variables..., [copy destinations...]).  As written by the
  reference, set/get records list the full-tensor variables 1-BASED (:218,242: ``newvars`` without -1) while copies
  list scalar variables 0-based (:271-272); this reader follows the writer.
* ``.cgs``  symbolic names: (type 1 constant | 2 function | 3 variable | 4 io, index, name length, description
  length, name, description).

What the reference's writer leaves out and the solver needs travels in extension records that its own readers skip:
opcodes the registry has no S-name for (saveScalarized.m:139-195 ``otherwise error``: I_heaviside, I_relu, I_cube, ...,
I_zero, I_componentwise) get function type 21000 + opcode, I_componentwise's C string is a string constant as
first input; shapes / names / the external flag of sets and gets go in the description field of their .cgs record;
the solver #define block (cmex2optimizeCS.m:303-331) is a .cgs record of type 5.  ``load_cg`` does not need any of
them for a reference-written file: names come from the gateway names, shapes from the subscripts, defines from the
``defines`` argument.
"""
from __future__ import annotations

import struct

import numpy as np

from . import instructions as I

C_DOUBLE, C_INT_VECTOR, C_INT_TENSOR, C_STRING = 10001, 20001, 20002, 30001
F_CONST, F_S2FT, F_FT2S, F_EXT0 = 0, 30001, 30002, 21000
IO_SET, IO_GET, IO_COPY = 1, 2, 3
SYM_CONST, SYM_FUN, SYM_VAR, SYM_IO, SYM_DEFINE = 1, 2, 3, 4, 5
# registry order of the scalar functions (CGregistration.m:67-130): type = 20001 + position
_S_NAMES = ["sum", "sumprod", "inv", "minus_inv_sqr", "div", "minus_dot", "minus_dot_div", "plus_minus_dot",
            "plus_sqr", "plus_minus_dot_div", "min", "min0", "max", "max0", "max_abs", "clp", "exp", "log", "cos",
            "minus_cos", "sin", "minus_sin", "abs", "sqrt", "Dsqrt", "DDsqrt", "sqr", "2times", "2", "atan", "Datan",
            "DDatan"]
TYPE_OF_OPCODE = {getattr(I, "I_" + nm): 20001 + k for k, nm in enumerate(_S_NAMES)}
OPCODE_OF_TYPE = {v: k for k, v in TYPE_OF_OPCODE.items()}


def function_type(opcode):
    return TYPE_OF_OPCODE.get(int(opcode), F_EXT0 + int(opcode))


class _Writer:
    def __init__(self, path, word):
        self.f = open(path, "wb")
        self.fmt = "<%dq" if word == 64 else "<%di"
        self.f.write(bytes([word]))

    def w(self, *ints):
        self.f.write(struct.pack(self.fmt % len(ints), *[int(x) for x in ints]))

    def raw(self, b):
        self.f.write(b)

    def close(self):
        self.f.close()


def save_cg(T, basename, word=64, extensions=True):
    """write basename.{cg,cgc,cgio,cgs} for Tape T.

    word=32, extensions=False reproduces what the reference's saveScalarized writes (int32 index words, no
    descriptions, no defines) -- such a file loads only with load_cg(..., defines=...) and cannot carry opcodes
    the registry has no name for; extensions="opcodes": the same, but unregistered opcodes are written with their
    extension type ids (a reference file of a tape the reference's writer would refuse)."""
    consts = []                       # (type, payload)
    nvar = [T.n]
    fg, fio, fs = (_Writer(basename + e, word) for e in (".cg", ".cgio", ".cgs"))

    def const_var(kind, payload):
        """constant + constant function node -> variable id (0-based)"""
        consts.append((kind, payload))
        v = nvar[0]
        nvar[0] += 1
        fg.w(F_CONST, len(consts) - 1, v)
        return v
    for i in range(1, T.n + 1):
        typ, par, ops = T.instr(i)
        if typ == I.I_set:
            continue
        if typ == I.I_load:
            consts.append((C_DOUBLE, struct.pack("<qqd", 1, 1, float(par[0]))))
            fg.w(F_CONST, len(consts) - 1, i - 1)
            continue
        ft = function_type(typ)
        if ft >= F_EXT0 and not extensions:
            raise ValueError("opcode %s has no registered scalar function (CGregistration.m)" % I.NAMES.get(typ, typ))
        if typ in (I.I_sum, I.I_sumprod):
            v = const_var(C_INT_VECTOR, struct.pack("<%dq" % len(par), *[int(p) for p in par]))
            fg.w(ft, 1 + len(ops), 1, v, *[o - 1 for o in ops], i - 1)
        elif typ == I.I_componentwise:
            s_ = bytes(int(p) for p in par)          # the C string, 0 = operand hole (instructionsTableUTHash.c:1333)
            v = const_var(C_STRING, s_)
            fg.w(ft, 1 + len(ops), 1, v, *[o - 1 for o in ops], i - 1)
        else:
            fg.w(ft, len(ops), 1, *[o - 1 for o in ops], i - 1)
    nio = [0]

    def sym(name, desc=""):
        if extensions is not True:
            desc = ""
        fs.w(SYM_IO, nio[0], len(name), len(desc))
        fs.raw(name.encode() + desc.encode())
        nio[0] += 1

    def subscripts(size, n):
        """full tensor: all subscripts, first dimension fastest (MATLAB order), as an int64 tensor [nd, numel]"""
        nd = max(1, len(size))
        sz = list(size) if len(size) else [1]
        idx = np.indices(sz[::-1]).reshape(nd, -1)[::-1] + 1 if n else np.zeros((nd, 0), dtype=np.int64)
        return struct.pack("<%dq" % (3 + idx.size), 2, nd, idx.shape[1], *idx.T.reshape(-1).tolist())
    for e in T.gets:                                            # saveScalarized.m:203-222
        newvars = []
        for o in e["outputs"]:
            v = const_var(C_INT_TENSOR, subscripts(o["size"], len(o["ins"])))
            nv = nvar[0]
            nvar[0] += 1
            fg.w(F_S2FT, 1 + len(o["ins"]), 1, v, *[i - 1 for i in o["ins"]], nv)
            newvars.append(nv + 1)                               # 1-based, as the reference writes them
        fio.w(IO_GET, len(newvars), *newvars)
        sym(e["name"], ";".join("%s|%s" % (o["name"], ",".join(str(s) for s in o["size"])) for o in e["outputs"]))
    for e in T.sets:                                            # :223-247
        nv = nvar[0]
        nvar[0] += 1
        fg.w(F_FT2S, 1, len(e["ins"]), nv, *[i - 1 for i in e["ins"]])
        fio.w(IO_SET, 1, nv + 1)
        sym(e["name"], "%s|%s|%d" % (e["varname"], ",".join(str(s) for s in e["size"]), int(e["external"])))
    for e in T.copies:                                          # :248-277
        fio.w(IO_COPY, len(e["src"]), *[i - 1 for i in e["src"]], *[i - 1 for i in e["dst"]])
        sym(e["name"])
    if extensions is True:
        for k, v in sorted(T.defines.items()):
            d = repr(float(v))
            fs.w(SYM_DEFINE, 0, len(k), len(d))
            fs.raw(k.encode() + d.encode())
    fc = _Writer(basename + ".cgc", word)
    for kind, payload in consts:
        fc.w(kind, len(payload))
        fc.raw(payload)
    for f in (fg, fio, fs, fc):
        f.close()


class _Reader:
    def __init__(self, path):
        self.d = open(path, "rb").read()
        if not self.d or self.d[0] not in (32, 64):
            raise ValueError("%s: unsupported magic %r (int32l / int64l index words expected)" % (path, self.d[:1]))
        self.ws = self.d[0] // 8
        self.fmt = "<%dq" if self.ws == 8 else "<%di"
        self.off = 1

    def more(self):
        return self.off < len(self.d)

    def ints(self, n):
        v = struct.unpack_from(self.fmt % n, self.d, self.off)
        self.off += self.ws * n
        return v

    def raw(self, n):
        b = self.d[self.off:self.off + n]
        self.off += n
        return b


_DOUBLES = {"gradTolerance", "equalTolerance", "desiredDualityGap", "addEye2HessianUtolerance", "alphaMin",
            "alphaMax", "muFactorAggressive", "muFactorConservative"}


def load_cg(basename, defines=None):
    """read the four files back -> Tape (dependency groups recomputed).  Accepts files written by the reference's
    saveScalarized (int32 words, registry type ids, no descriptions; pass the solver ``defines``) and by save_cg."""
    from .csparse import Tape
    r = _Reader(basename + ".cgc")
    consts = []
    while r.more():
        kind, ln = r.ints(2)
        consts.append((kind, r.raw(ln)))

    def const_ints(k):
        kind, b = consts[k]
        v = struct.unpack("<%dq" % (len(b) // 8), b)
        return list(v) if kind == C_INT_VECTOR else v      # tensors: (nd, size..., values)
    r = _Reader(basename + ".cg")
    recs, const_of_var, s2ft, ft2s = {}, {}, {}, {}
    nvar = 0
    while r.more():
        (typ,) = r.ints(1)
        if typ == F_CONST:
            k, out = r.ints(2)
            kind, b = consts[k]
            if kind == C_DOUBLE:
                nd = struct.unpack_from("<q", b, 0)[0]
                vals = struct.unpack_from("<%dd" % ((len(b) - 8 * (1 + nd)) // 8), b, 8 * (1 + nd))
                if len(vals) == 1:
                    recs[out + 1] = (I.I_load, [vals[0]], [])
            const_of_var[out] = k
            nvar = max(nvar, out + 1)
            continue
        nin, nout = r.ints(2)
        ins, outs = list(r.ints(nin)), list(r.ints(nout))
        nvar = max([nvar] + [v + 1 for v in ins + outs])
        if typ == F_S2FT:
            s2ft[outs[0]] = (const_of_var[ins[0]], [v + 1 for v in ins[1:]])
            continue
        if typ == F_FT2S:
            ft2s[ins[0]] = [v + 1 for v in outs]
            continue
        op = OPCODE_OF_TYPE.get(typ, typ - F_EXT0 if typ >= F_EXT0 else None)
        if op is None or op not in I.NAMES:
            raise ValueError("unknown function type %d in %s.cg" % (typ, basename))
        par = []
        if op in (I.I_sum, I.I_sumprod):
            par = [float(p) for p in const_ints(const_of_var[ins[0]])]
            ins = ins[1:]
        elif op == I.I_componentwise:
            par = [float(c) for c in consts[const_of_var[ins[0]]][1]]
            ins = ins[1:]
        recs[outs[0] + 1] = (op, par, [v + 1 for v in ins])
    # symbols
    r = _Reader(basename + ".cgs")
    names, defs = [], {}
    while r.more():
        kind, idx, ln, ld = r.ints(4)
        nm, ds = r.raw(ln).decode(), r.raw(ld).decode()
        if kind == SYM_IO:
            names.append((nm, ds))
        elif kind == SYM_DEFINE:
            v = float(ds)
            defs[nm] = v if nm in _DOUBLES or not v.is_integer() else int(v)
    if defines is not None:
        defs.update(defines)
    # gateway functions
    r = _Reader(basename + ".cgio")
    sets, gets, copies, q = [], [], [], 0
    tensor_vars = set(s2ft) | set(ft2s)
    while r.more():
        kind, n = r.ints(2)
        v = list(r.ints(n))
        nm, ds = names[q] if q < len(names) else ("io%d" % q, "")
        q += 1
        if kind == IO_COPY:
            d = list(r.ints(n))
            copies.append(dict(name=nm, src=np.array(v, dtype=np.int64) + 1, dst=np.array(d, dtype=np.int64) + 1))
            continue
        # the reference writes these ids 1-based; tolerate 0-based files too
        tv = [x - 1 if (x - 1) in tensor_vars else x for x in v]
        if kind == IO_GET:
            descs = ds.split(";") if ds else [""] * len(tv)
            outs = []
            for k, t_ in enumerate(tv):
                ck, ins = s2ft[t_]
                sub = const_ints(ck)
                nd = sub[0]
                dims, vals = sub[1:1 + nd], np.array(sub[1 + nd:], dtype=np.int64)
                if descs[k]:
                    oname, sz = descs[k].split("|")
                    size = tuple(int(s) for s in sz.split(",") if s)
                else:                  # reference file: shape = largest subscript per dimension ([nd x numel] tensor)
                    oname = "y%d" % (k + 1)
                    sb = vals.reshape(dims[::-1]).T if len(dims) == 2 else vals.reshape(1, -1)
                    size = tuple(int(x) for x in sb.max(axis=1)) if sb.size else (0,)
                    if all(s == 1 for s in size):
                        size = ()
                outs.append(dict(name=oname, size=size, ins=np.array(ins, dtype=np.int64)))
            gets.append(dict(name=nm, outputs=outs))
        else:
            ins = [i for t_ in tv for i in ft2s[t_]]
            if ds:
                vname, sz, ext = ds.split("|")
                size, ext = tuple(int(s) for s in sz.split(",") if s), bool(int(ext))
            else:                      # reference file: <cls>_set_<name> is a user-facing set, xxx__ an internal one
                vname = nm.split("_set_", 1)[1] if "_set_" in nm else nm
                size, ext = ((len(ins),) if len(ins) != 1 else ()), not nm.endswith("__")
            sets.append(dict(name=nm, varname=vname, size=size, ins=np.array(ins, dtype=np.int64), external=ext))
    ninstr = max([max(recs, default=0)] + [int(np.max(e["ins"])) for e in sets if len(e["ins"])] +
                 [int(max(e["src"].max(), e["dst"].max())) for e in copies if len(e["src"])])
    # instruction table: variables not produced by any function node are I_set
    types = np.full(ninstr, I.I_set, dtype=np.int32)
    opStart, parStart, ops, pars = [0], [0], [], []
    for i in range(1, ninstr + 1):
        if i in recs:
            t, p, o = recs[i]
            types[i - 1] = t
            ops += o
            pars += p
        opStart.append(len(ops))
        parStart.append(len(pars))
    return Tape.from_instructions(types, np.array(opStart, dtype=np.int64), np.array(ops, dtype=np.int64),
                                  np.array(parStart, dtype=np.int64), np.array(pars, dtype=np.float64), sets,
                                  gets, copies, defs)
