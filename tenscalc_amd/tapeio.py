"""Binary tape container shared by the Python producer, the C oracle and the C++ runtime.

A flat list of named records (name[32], dtype, count, payload padded to 8 bytes), in the
spirit of the reference's computation-graph export (``@csparse/saveScalarized.m:103-321``,
doc/computationgraphs.tex:84-220) but carrying what that format lacks: dependency groups,
entry-point group lists and the solver ``#define`` block.
"""
from __future__ import annotations

import struct

import numpy as np

MAGIC = b"TCTAPE1\0"
_I64, _F64, _BYTES = 0, 1, 2


def _rec(name, arr):
    nm = name.encode()[:31].ljust(32, b"\0")
    if isinstance(arr, (bytes, str)):
        b = arr.encode() if isinstance(arr, str) else arr
        pad = (-len(b)) % 8
        return nm + struct.pack("<qq", _BYTES, len(b)) + b + b"\0" * pad
    a = np.asarray(arr)
    if a.dtype.kind == "f":
        a = np.ascontiguousarray(a, dtype="<f8").reshape(-1)
        return nm + struct.pack("<qq", _F64, a.size) + a.tobytes()
    a = np.ascontiguousarray(a, dtype="<i8").reshape(-1)
    return nm + struct.pack("<qq", _I64, a.size) + a.tobytes()


def tape_records(T):
    r = {}
    dn = sorted(T.defines)
    r["defines_names"] = "\n".join(dn)
    r["defines_values"] = np.array([float(T.defines[k]) for k in dn], dtype=np.float64)
    r["types"] = T.types
    r["opStart"] = T.opStart
    r["ops"] = T.ops
    r["parStart"] = T.parStart
    r["pars"] = T.pars
    r["group"] = T.group
    r["slot"] = T.slot
    r["nSlots"] = [T.nSlots]
    r["nGroups"] = [T.nGroups]
    gs = np.cumsum([0] + [len(g) for g in T.groups])
    r["grpStart"] = gs
    r["grpInstr"] = np.concatenate(T.groups) if T.groups else np.zeros(0, dtype=np.int64)
    r["grpFlops"] = T.flops.sum(axis=1)
    names, kinds = [], []
    k = 0
    for e in T.sets:
        names.append(e["name"])
        kinds.append(0)
        r[f"e{k}_a"] = e["ins"]
        r[f"e{k}_dims"] = list(e["size"])
        r[f"e{k}_children"] = e["children"]
        r[f"e{k}_ext"] = [int(e["external"])]
        r[f"e{k}_outnames"] = e["varname"]
        k += 1
    for e in T.gets:
        names.append(e["name"])
        kinds.append(1)
        r[f"e{k}_a"] = np.concatenate([o["ins"] for o in e["outputs"]])
        r[f"e{k}_b"] = np.cumsum([0] + [len(o["ins"]) for o in e["outputs"]])
        dims = []
        for o in e["outputs"]:
            dims += [len(o["size"])] + list(o["size"])
        r[f"e{k}_dims"] = dims
        r[f"e{k}_parents"] = e["parents"]
        r[f"e{k}_outnames"] = "\n".join(o["name"] for o in e["outputs"])
        k += 1
    for e in T.copies:
        names.append(e["name"])
        kinds.append(2)
        r[f"e{k}_a"] = e["dst"]
        r[f"e{k}_b"] = e["src"]
        r[f"e{k}_parents"] = e["parents"]
        r[f"e{k}_children"] = e["children"]
        k += 1
    r["entry_names"] = "\n".join(names)
    r["entry_kind"] = kinds
    return r


def save_tape(T, path):
    with open(path, "wb") as f:
        f.write(MAGIC)
        for name, arr in tape_records(T).items():
            f.write(_rec(name, arr))
        f.write(_rec("END", [0]))


def load_records(path):
    """Read back the raw records (used by tests and by the .cg importer)."""
    out = {}
    with open(path, "rb") as f:
        data = f.read()
    if data[:8] != MAGIC:
        raise ValueError("not a tape file")
    off = 8
    while off < len(data):
        name = data[off:off + 32].split(b"\0", 1)[0].decode()
        dt, cnt = struct.unpack_from("<qq", data, off + 32)
        off += 48
        if dt == _BYTES:
            out[name] = data[off:off + cnt].decode()
            off += cnt + ((-cnt) % 8)
        elif dt == _F64:
            out[name] = np.frombuffer(data, dtype="<f8", count=cnt, offset=off).copy()
            off += 8 * cnt
        else:
            out[name] = np.frombuffer(data, dtype="<i8", count=cnt, offset=off).copy()
            off += 8 * cnt
        if name == "END":
            break
    return out
