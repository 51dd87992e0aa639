"""Scalar instruction ISA + ctypes binding of the native instruction table.

Opcode numbering follows the reference so that tapes are interchangeable with its
computation-graph export (/root/reference/lib/instructionTypes.m:8-117,
lib/csparse/instructionsTableTypes.h:4-58).  Semantics of every opcode are those
printed by ``writeCinstructionsC`` (lib/csparse/instructionsTableUTHash.c:950-1599).
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

# C-level opcodes (UMFPACK atomics 51-54 are out of scope: SURVEY.md section 2)
I_set = 1
I_load = 2
I_sum = 3
I_sumprod = 4
I_inv = 5
I_minus_inv_sqr = 6
I_2_inv_cube = 7
I_div = 8
I_minus_dot = 9
I_minus_dot_div = 10
I_plus_minus_dot = 11
I_plus_sqr = 12
I_plus_abs = 13
I_plus_minus_dot_div = 14
I_min = 15
I_min0 = 16
I_max = 17
I_max0 = 18
I_max_abs = 19
I_clp = 20
I_exp = 21
I_log = 22
I_cos = 23
I_minus_cos = 24
I_sin = 25
I_minus_sin = 26
I_round = 27
I_ceil = 28
I_floor = 29
I_abs = 30
I_sign = 31
I_sqrt = 32
I_Dsqrt = 33
I_DDsqrt = 34
I_sqr = 35
I_2times = 36
I_2 = 37
I_cube = 38
I_3sqr = 39
I_6times = 40
I_atan = 41
I_Datan = 42
I_DDatan = 43
I_srelu = 44
I_dsrelu = 45
I_ddsrelu = 46
I_relu = 47
I_heaviside = 48
I_zero = 49
I_componentwise = 50

NAMES = {v: k for k, v in list(globals().items()) if k.startswith("I_") and isinstance(v, int)}

# compose() function name -> opcode (computeScalarInstructions.m:366-410)
COMPOSE_OPCODE = {
    "log": I_log, "inv": I_inv, "minus_inv_sqr": I_minus_inv_sqr, "2_inv_cube": I_2_inv_cube,
    "exp": I_exp, "sin": I_sin, "minus_sin": I_minus_sin, "cos": I_cos, "minus_cos": I_minus_cos,
    "round": I_round, "ceil": I_ceil, "floor": I_floor, "abs": I_abs, "sign": I_sign,
    "sqrt": I_sqrt, "Dsqrt": I_Dsqrt, "DDsqrt": I_DDsqrt, "sqr": I_sqr, "2times": I_2times,
    "2": I_2, "cube": I_cube, "3sqr": I_3sqr, "6times": I_6times, "atan": I_atan,
    "Datan": I_Datan, "DDatan": I_DDatan, "srelu": I_srelu, "dsrelu": I_dsrelu,
    "ddsrelu": I_ddsrelu, "relu": I_relu, "heaviside": I_heaviside,
}
# functions f with f(0)==0: structural zeros of the operand stay structural zeros
# (sparsity_compose.m vs sparsity_compose_full.m)
COMPOSE_KEEPS_ZERO = {
    "sin", "minus_sin", "round", "ceil", "floor", "abs", "sqrt", "sqr", "2times", "cube",
    "3sqr", "6times", "atan", "DDatan", "relu", "heaviside",
}

# I_componentwise: the data-driven evaluators (oracle, interp, ldsm) cannot run arbitrary C text; they accept the
# strings that are a plain call of one of these functions on the operand and run the matching opcode instead
COMPONENTWISE_CALLS = {"fabs": I_abs, "abs": I_abs, "sqrt": I_sqrt, "exp": I_exp, "log": I_log, "sin": I_sin,
                       "cos": I_cos, "atan": I_atan, "round": I_round, "ceil": I_ceil, "floor": I_floor}


def componentwise_text(par):
    """parameters of an I_componentwise instruction -> (C text with %s for the operand hole)"""
    return "".join("%s" if int(p) == 0 else chr(int(p)) for p in par)


def componentwise_opcode(par):
    """opcode equivalent to the C text, or None"""
    import re
    m = re.fullmatch(r"\s*([A-Za-z_][A-Za-z0-9_]*)\s*\(\s*%s\s*\)\s*", componentwise_text(par))
    return COMPONENTWISE_CALLS.get(m.group(1)) if m else None


# flop-count classes (instructionsTableTypes.h:108-126)
P_nsum, P_nprod, P_ndiv, P_nif, P_nclp, P_nabs, P_nround, P_nceil, P_nfloor, P_nsign, \
    P_nsqrt, P_npow, P_ntrig, P_nlog, P_nexp, P_cpwise, P_numfpack = range(17)
P_NAMES = ["nsum", "nprod", "ndiv", "nif", "nclp", "nabs", "nround", "nceil", "nfloor",
           "nsign", "nsqrt", "npow", "ntrig", "nlog", "nexp", "cpwise", "numfpack"]

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        # TC_SANITIZE: the AddressSanitizer / UBSan build of the table (scripts/sanitize_cpu.sh)
        path = os.path.join(_HERE, "libtcitable_san.so" if os.environ.get("TC_SANITIZE") else "libtcitable.so")
        if not os.path.exists(path):
            raise RuntimeError(
                f"{path} missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        L = ctypes.CDLL(path)
        i64, i32, dp, ip = ctypes.c_int64, ctypes.c_int32, ctypes.POINTER(ctypes.c_double), \
            ctypes.POINTER(ctypes.c_int64)
        L.tcit_create.restype = ctypes.c_void_p
        L.tcit_destroy.argtypes = [ctypes.c_void_p]
        for f in (L.tcit_append, L.tcit_append_unique):
            f.restype = i64
            f.argtypes = [ctypes.c_void_p, i32, i64, dp, i64, ip]
        L.tcit_append_many.restype = ctypes.c_int
        L.tcit_append_many.argtypes = [ctypes.c_void_p, i32, i64, i64, dp, i64, ip, i32, ip]
        L.tcit_height.restype = i64
        L.tcit_height.argtypes = [ctypes.c_void_p]
        L.tcit_total_operands.restype = i64
        L.tcit_total_operands.argtypes = [ctypes.c_void_p]
        L.tcit_total_parameters.restype = i64
        L.tcit_total_parameters.argtypes = [ctypes.c_void_p]
        L.tcit_export.argtypes = [ctypes.c_void_p, ctypes.POINTER(i32), ip, ip, ip, dp]
        L.tcit_dependencies.argtypes = [ctypes.c_void_p, ip, ip]
        L.tcit_find_by_type.argtypes = [ctypes.c_void_p, i32, ctypes.POINTER(ctypes.c_int8)]
        L.tcit_get.restype = ctypes.c_int
        L.tcit_get.argtypes = [ctypes.c_void_p, i64, ctypes.POINTER(i32), ip,
                               ctypes.POINTER(dp), ip, ctypes.POINTER(ip)]
        _LIB = L
    return _LIB


def _dp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def _ip(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))


_EMPTY_D = np.zeros(1, dtype=np.float64)
_EMPTY_I = np.zeros(1, dtype=np.int64)


class InstructionsTable:
    """Handle on one native table (the reference has a single global one,
    csparse.m:255-259 ``loadlibrary('instructionsTable')``)."""

    def __init__(self):
        self._L = _lib()
        self._h = ctypes.c_void_p(self._L.tcit_create())

    def __del__(self):
        try:
            if self._h:
                self._L.tcit_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def __len__(self):
        return int(self._L.tcit_height(self._h))

    def append(self, typ, parameters=(), operands=(), unique=True):
        p = np.ascontiguousarray(parameters, dtype=np.float64).reshape(-1)
        o = np.ascontiguousarray(operands, dtype=np.int64).reshape(-1)
        f = self._L.tcit_append_unique if unique else self._L.tcit_append
        r = f(self._h, int(typ), p.size, _dp(p if p.size else _EMPTY_D), o.size,
              _ip(o if o.size else _EMPTY_I))
        if r < 0:
            raise RuntimeError(f"instruction table rejected {NAMES.get(typ, typ)} ops={o}")
        return int(r)

    def append_many(self, typ, parameters, operands, unique=True):
        """``operands`` [n, nOps] and ``parameters`` [n, nPars] (or None) -> indices [n]."""
        o = np.ascontiguousarray(operands, dtype=np.int64)
        n = o.shape[0]
        if parameters is None:
            p = np.zeros((n, 0))
        else:
            p = np.ascontiguousarray(parameters, dtype=np.float64).reshape(n, -1)
        out = np.empty(n, dtype=np.int64)
        if n == 0:
            return out
        nO = o.shape[1] if o.ndim == 2 else 0
        rc = self._L.tcit_append_many(self._h, int(typ), n, p.shape[1],
                                      _dp(p if p.size else _EMPTY_D), nO,
                                      _ip(o if o.size else _EMPTY_I), 1 if unique else 0, _ip(out))
        if rc:
            raise RuntimeError("instruction table rejected a batch append")
        return out

    def export(self):
        """(types[n], opStart[n+1], ops, parStart[n+1], pars) as numpy arrays."""
        n = len(self)
        types = np.empty(n, dtype=np.int32)
        opStart = np.empty(n + 1, dtype=np.int64)
        parStart = np.empty(n + 1, dtype=np.int64)
        ops = np.empty(max(1, self._L.tcit_total_operands(self._h)), dtype=np.int64)
        pars = np.empty(max(1, self._L.tcit_total_parameters(self._h)), dtype=np.float64)
        self._L.tcit_export(self._h, types.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                            _ip(opStart), _ip(ops), _ip(parStart), _dp(pars))
        return types, opStart, ops[:opStart[n]], parStart, pars[:parStart[n]]

    def dependencies(self):
        m = int(self._L.tcit_total_operands(self._h))
        c = np.empty(max(1, m), dtype=np.int64)
        p = np.empty(max(1, m), dtype=np.int64)
        self._L.tcit_dependencies(self._h, _ip(c), _ip(p))
        return c[:m], p[:m]

    def find_by_type(self, typ):
        n = len(self)
        out = np.zeros(max(1, n), dtype=np.int8)
        self._L.tcit_find_by_type(self._h, int(typ), out.ctypes.data_as(ctypes.POINTER(ctypes.c_int8)))
        return out[:n].astype(bool)

    def get(self, index):
        t = ctypes.c_int32()
        nP, nO = ctypes.c_int64(), ctypes.c_int64()
        pp = ctypes.POINTER(ctypes.c_double)()
        po = ctypes.POINTER(ctypes.c_int64)()
        rc = self._L.tcit_get(self._h, int(index), ctypes.byref(t), ctypes.byref(nP),
                              ctypes.byref(pp), ctypes.byref(nO), ctypes.byref(po))
        if rc:
            raise IndexError(index)
        pars = np.array([pp[i] for i in range(nP.value)], dtype=np.float64)
        ops = np.array([po[i] for i in range(nO.value)], dtype=np.int64)
        return t.value, pars, ops
