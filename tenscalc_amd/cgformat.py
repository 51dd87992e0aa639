"""Computation-graph files (.cg/.cgc/.cgio/.cgs) for scalarized tapes.

File layout as documented in /root/reference/doc/computationgraphs.tex:84-220 and written by
``@csparse/saveScalarized.m:103-321``: four files, each starting with a one-byte magic (64 =
little-endian int64 indices) followed by variable-length records.

* ``.cgc``  constants: (type, byte length, data); ``I_load`` values and the parameter vectors of
  ``I_sum``/``I_sumprod`` are constants (saveScalarized.m:176-190).
* ``.cg``   one record per function node: (type, nIn, nOut, inputs..., outputs...), 0-based
  variable ids = instruction index - 1; type 0 = constant node (constant id, output variable).
* ``.cgio`` gateway functions: (1 set | 2 get | 3 copy, n, variables..., [copy destinations...]).
* ``.cgs``  symbolic names: (4 = io function, index, name length, description length, name, desc).

The reference's registry file gives every scalar function the same type id (CGregistration.json:
all ``S*`` entries are 20002) and does not register sqrt/componentwise-family opcodes
(saveScalarized.m:139-195), so its own files are ambiguous.  We keep the documented container and
register scalar function type = 20000 + opcode (lib/csparse/instructionsTableTypes.h:4-58), plus
two record kinds the solver needs and the format lacks: the external/internal flag and shape of
every set (in the .cgs description) and the solver #define block (a .cgs record of type 5).
"""
from __future__ import annotations

import struct

import numpy as np

from . import instructions as I

MAGIC = 64
C_DOUBLE, C_INT_VECTOR, C_STRING = 10001, 10003, 10004
F_CONST, F_SCALAR0 = 0, 20000
IO_SET, IO_GET, IO_COPY = 1, 2, 3
SYM_CONST, SYM_FUN, SYM_VAR, SYM_IO, SYM_DEFINE = 1, 2, 3, 4, 5


def _w(f, *ints):
    f.write(struct.pack("<%dq" % len(ints), *[int(x) for x in ints]))


def save_cg(T, basename):
    """write basename.{cg,cgc,cgio,cgs} for Tape T"""
    consts = []

    def add_const(kind, payload):
        consts.append((kind, payload))
        return len(consts) - 1
    with open(basename + ".cg", "wb") as fg:
        fg.write(bytes([MAGIC]))
        for i in range(1, T.n + 1):
            typ, par, ops = T.instr(i)
            if typ == I.I_set:
                continue
            if typ == I.I_load:
                k = add_const(C_DOUBLE, struct.pack("<qqd", 1, 1, float(par[0])))
                _w(fg, F_CONST, k, i - 1)
            elif typ in (I.I_sum, I.I_sumprod):
                k = add_const(C_INT_VECTOR, struct.pack("<%dq" % len(par), *[int(p) for p in par]))
                # the parameter constant is the first input, itself a variable produced by a constant node
                _w(fg, F_SCALAR0 + typ, 1 + len(ops), 1, -(k + 1), *[o - 1 for o in ops], i - 1)
            else:
                _w(fg, F_SCALAR0 + typ, len(ops), 1, *[o - 1 for o in ops], i - 1)
    with open(basename + ".cgio", "wb") as fio, open(basename + ".cgs", "wb") as fs:
        fio.write(bytes([MAGIC]))
        fs.write(bytes([MAGIC]))
        nio = 0

        def sym(name, desc=""):
            nonlocal nio
            _w(fs, SYM_IO, nio, len(name), len(desc))
            fs.write(name.encode() + desc.encode())
            nio += 1
        for e in T.gets:
            for o in e["outputs"]:
                _w(fio, IO_GET, len(o["ins"]), *[i - 1 for i in o["ins"]])
                sym(e["name"], "%s|%s" % (o["name"], ",".join(str(s) for s in o["size"])))
        for e in T.sets:
            _w(fio, IO_SET, len(e["ins"]), *[i - 1 for i in e["ins"]])
            sym(e["name"], "%s|%s|%d" % (e["varname"], ",".join(str(s) for s in e["size"]), int(e["external"])))
        for e in T.copies:
            _w(fio, IO_COPY, len(e["src"]), *[i - 1 for i in e["src"]], *[i - 1 for i in e["dst"]])
            sym(e["name"])
        for k, v in sorted(T.defines.items()):
            d = repr(float(v))
            _w(fs, SYM_DEFINE, 0, len(k), len(d))
            fs.write(k.encode() + d.encode())
    with open(basename + ".cgc", "wb") as fc:
        fc.write(bytes([MAGIC]))
        for kind, payload in consts:
            _w(fc, kind, len(payload))
            fc.write(payload)


def _ints(data, off, n):
    return struct.unpack_from("<%dq" % n, data, off), off + 8 * n


def load_cg(basename):
    """read the four files back -> Tape (dependency groups recomputed)"""
    from .csparse import Tape
    data = open(basename + ".cgc", "rb").read()
    assert data[0] == MAGIC, "only little-endian int64 indices are supported"
    consts, off = [], 1
    while off < len(data):
        (kind, ln), off = _ints(data, off, 2)
        consts.append((kind, data[off:off + ln]))
        off += ln
    data = open(basename + ".cg", "rb").read()
    assert data[0] == MAGIC
    recs, off, nvar = {}, 1, 0
    while off < len(data):
        (typ,), off = _ints(data, off, 1)
        if typ == F_CONST:
            (k, out), off = _ints(data, off, 2)
            val = struct.unpack_from("<d", consts[k][1], 16)[0]
            recs[out + 1] = (I.I_load, [val], [])
        else:
            (nin, nout), off = _ints(data, off, 2)
            ins, off = _ints(data, off, nin)
            (out,), off = _ints(data, off, nout)
            op = typ - F_SCALAR0
            par = []
            ins = list(ins)
            if op in (I.I_sum, I.I_sumprod):
                k = -ins[0] - 1
                par = list(struct.unpack("<%dq" % (len(consts[k][1]) // 8), consts[k][1]))
                ins = ins[1:]
            recs[out + 1] = (op, [float(p) for p in par], [i + 1 for i in ins])
        nvar = max(nvar, max(recs))
    data = open(basename + ".cgio", "rb").read()
    sdata = open(basename + ".cgs", "rb").read()
    names, defines, soff = [], {}, 1
    while soff < len(sdata):
        (kind, idx, ln, ld), soff = _ints(sdata, soff, 4)
        nm = sdata[soff:soff + ln].decode()
        ds = sdata[soff + ln:soff + ln + ld].decode()
        soff += ln + ld
        if kind == SYM_IO:
            names.append((nm, ds))
        elif kind == SYM_DEFINE:
            v = float(ds)
            defines[nm] = int(v) if v.is_integer() and "olerance" not in nm and "Factor" not in nm \
                and nm not in ("alphaMin", "alphaMax", "desiredDualityGap") else v
    sets, gets, copies, off, q = [], {}, [], 1, 0
    get_order = []
    while off < len(data):
        (kind, n), off = _ints(data, off, 2)
        v, off = _ints(data, off, n)
        v = np.array(v, dtype=np.int64) + 1
        nm, ds = names[q]
        q += 1
        nvar = max(nvar, int(v.max()) if v.size else 0)
        if kind == IO_GET:
            oname, sz = ds.split("|")
            size = tuple(int(s) for s in sz.split(",") if s)
            if nm not in gets:
                gets[nm] = dict(name=nm, outputs=[])
                get_order.append(nm)
            gets[nm]["outputs"].append(dict(name=oname, size=size, ins=v))
        elif kind == IO_SET:
            vname, sz, ext = ds.split("|")
            sets.append(dict(name=nm, varname=vname, size=tuple(int(s) for s in sz.split(",") if s), ins=v,
                             external=bool(int(ext))))
        else:
            d, off = _ints(data, off, n)
            copies.append(dict(name=nm, src=v, dst=np.array(d, dtype=np.int64) + 1))
            nvar = max(nvar, max(d) + 1)
    # instruction table: variables not produced by any function node are I_set
    types = np.full(nvar, I.I_set, dtype=np.int32)
    opStart, parStart, ops, pars = [0], [0], [], []
    for i in range(1, nvar + 1):
        if i in recs:
            t, p, o = recs[i]
            types[i - 1] = t
            ops += o
            pars += p
        opStart.append(len(ops))
        parStart.append(len(pars))
    return Tape.from_instructions(types, np.array(opStart, dtype=np.int64), np.array(ops, dtype=np.int64),
                                  np.array(parStart, dtype=np.int64), np.array(pars, dtype=np.float64), sets,
                                  [gets[k] for k in get_order], copies, defines)
