"""The reference's "typical values" files: ``<cls>_WW.subscripts`` / ``<cls>_WW.values``.

``cmex2optimizeCS(..., allowSave=true)`` makes the generated solver able to write the values of its Newton matrix WW to a
file (``saveWW__``, lib/ipmPD_CSsolver.c:330-333,561-566; file format in @csparse/writeCfunctionpergroup.m:879-918), and
a later code generation uses them to choose the pivot order of the symbolic factorisation
(@csparse/sparsity_lu.m:36-57: MATLAB's ``lu`` on the typical matrix instead of on random numbers; sparsity_ldl.m:40-62).
This module reads and writes that pair of files (layout as @csparse/private/loadCSparse.m reads it):

    .subscripts   int64 magic | int32 nDim | int64 osize[nDim] | int64 nnz | int64 subscripts[nDim][nnz]  (1-based, column-major)
    .values       int64 magic | double values[nnz]

Batched back-end: ``HipSolver.saveWW(filename, instance)`` (runtime.py) evaluates the class's ``saveWW__`` entry point right
after the initialisation sequence -- the Newton matrix of the FIRST iteration, where the reference writes the one of
iteration ``saveIter`` -- for one instance of the batch and writes the ``.values`` file; the ``.subscripts`` file is written
at code-generation time, as in the reference."""
from __future__ import annotations

import os
import struct
import time

import numpy as np


def new_magic():
    """ns since 2000-01-01, csparse.m:590"""
    return int((time.time() - 946684800.0) * 1e9)


def write_subscripts(path, osize, subs, magic):
    """subs: [nnz, nDim] 0-based (this package's convention) -> the reference's 1-based [nDim][nnz] layout"""
    subs = np.asarray(subs, dtype=np.int64).reshape(-1, len(osize))
    with open(path, "wb") as f:
        f.write(struct.pack("<q", int(magic)))
        f.write(struct.pack("<i", len(osize)))
        f.write(np.asarray(osize, dtype="<i8").tobytes())
        f.write(struct.pack("<q", subs.shape[0]))
        f.write(np.ascontiguousarray((subs + 1).astype("<i8")).tobytes())      # [nnz][nDim] C-order == [nDim][nnz] column-major
    return path


def write_values(path, values, magic):
    with open(path, "wb") as f:
        f.write(struct.pack("<q", int(magic)))
        f.write(np.asarray(values, dtype="<f8").reshape(-1).tobytes())
    return path


def read_magic(path):
    with open(path, "rb") as f:
        return struct.unpack("<q", f.read(8))[0]


def read_subscripts(path):
    """-> (magic, osize tuple, subs [nnz, nDim] 0-based)"""
    a = open(path, "rb").read()
    magic, = struct.unpack_from("<q", a, 0)
    nd, = struct.unpack_from("<i", a, 8)
    osize = np.frombuffer(a, dtype="<i8", count=nd, offset=12)
    nnz, = struct.unpack_from("<q", a, 12 + 8 * nd)
    subs = np.frombuffer(a, dtype="<i8", count=nd * nnz, offset=20 + 8 * nd).reshape(nnz, nd) - 1
    return magic, tuple(int(x) for x in osize), subs.copy()


def load(path_subscripts, path_values, ignoremagic=False):
    """-> (osize tuple, subs [nnz, nDim] 0-based, values [nnz]) or None when a file is missing, the magic numbers differ
    or the sizes do not match (loadCSparse.m returns NaNs in those cases and the caller falls back to random values)"""
    if not (os.path.exists(path_subscripts) and os.path.exists(path_values)):
        return None
    a, b = open(path_subscripts, "rb").read(), open(path_values, "rb").read()
    if len(a) < 20 or len(b) < 8:
        return None
    m1, = struct.unpack_from("<q", a, 0)
    m2, = struct.unpack_from("<q", b, 0)
    if not ignoremagic and m1 != m2:
        return None
    nd, = struct.unpack_from("<i", a, 8)
    osize = np.frombuffer(a, dtype="<i8", count=nd, offset=12)
    nnz, = struct.unpack_from("<q", a, 12 + 8 * nd)
    subs = np.frombuffer(a, dtype="<i8", count=nd * nnz, offset=20 + 8 * nd).reshape(nnz, nd) - 1
    vals = np.frombuffer(b, dtype="<f8", offset=8)
    if vals.size != nnz:
        return None
    return tuple(int(x) for x in osize), subs.copy(), vals.copy()
