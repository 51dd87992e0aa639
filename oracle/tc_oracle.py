"""ORACLE - test infrastructure only.  ctypes face of oracle/tc_oracle.c (+ the optional
reference-loop build in oracle/_ref/)."""
import ctypes

import numpy as np

from . import build as _build

_dp = ctypes.POINTER(ctypes.c_double)
_i32p = ctypes.POINTER(ctypes.c_int32)


def _bind(L):
    L.tco_load.restype = ctypes.c_void_p
    L.tco_load.argtypes = [ctypes.c_char_p]
    L.tco_free.argtypes = [ctypes.c_void_p]
    L.tco_entry.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
    L.tco_entry_size.restype = ctypes.c_int64
    L.tco_entry_size.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.tco_set.argtypes = [ctypes.c_void_p, ctypes.c_int, _dp]
    L.tco_get_flat.argtypes = [ctypes.c_void_p, ctypes.c_int, _dp]
    L.tco_copy.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.tco_solve.argtypes = [ctypes.c_void_p, ctypes.c_double, ctypes.c_int32, ctypes.c_int32, _dp,
                            _i32p, _i32p]
    L.tco_solve_batch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int),
                                  ctypes.POINTER(_dp), ctypes.c_int, _dp, _i32p, _i32p,
                                  ctypes.c_int64, ctypes.c_double, ctypes.c_int32, _dp]
    L.tco_solve_eq.argtypes = L.tco_solve.argtypes
    L.tco_solve_eq_batch.argtypes = L.tco_solve_batch.argtypes
    L.tco_define.restype = ctypes.c_double
    L.tco_define.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_double]
    L.tco_value.restype = ctypes.c_double
    L.tco_value.argtypes = [ctypes.c_void_p, ctypes.c_int64]
    for f in ("tco_flops", "tco_group_evals", "tco_minF_trials"):
        getattr(L, f).restype = ctypes.c_int64
        getattr(L, f).argtypes = [ctypes.c_void_p]
    L.tco_reset_counters.argtypes = [ctypes.c_void_p]
    return L


class Oracle:
    """One problem instance at a time, exactly like the reference's generated library."""

    def __init__(self, tape_path, lib=None):
        self.L = _bind(ctypes.CDLL(lib or _build.build_oracle()))
        self.h = ctypes.c_void_p(self.L.tco_load(tape_path.encode()))
        if not self.h:
            raise RuntimeError("cannot load tape " + tape_path)
        # which Newton loop the tape was generated for: the equilibrium solvers' #define block has nZ
        # (cmex2equilibriumLatentCS.m:418), the optimisation solvers' has not
        self.eq = self.L.tco_define(self.h, b"nZ", -1.0) >= 0

    def entry(self, name):
        e = self.L.tco_entry(self.h, name.encode())
        if e < 0:
            raise KeyError(name)
        return e

    def size(self, name):
        return int(self.L.tco_entry_size(self.h, self.entry(name)))

    def set(self, name, value):
        v = np.ascontiguousarray(np.asarray(value, dtype=np.float64).reshape(-1, order="F"))
        if v.size != self.size(name):
            raise ValueError("%s expects %d values" % (name, self.size(name)))
        self.L.tco_set(self.h, self.entry(name), v.ctypes.data_as(_dp))

    def get(self, name):
        out = np.empty(self.size(name))
        self.L.tco_get_flat(self.h, self.entry(name), out.ctypes.data_as(_dp))
        return out

    def copy(self, name):
        self.L.tco_copy(self.h, self.entry(name))

    def solve(self, mu0=1.0, maxIter=200, saveIter=-1, addEye2Hessian=(1e-9, 1e-9)):
        ae = np.array(addEye2Hessian, dtype=np.float64)
        st, it = ctypes.c_int32(0), ctypes.c_int32(0)
        (self.L.tco_solve_eq if self.eq else self.L.tco_solve)(self.h, mu0, maxIter, saveIter, ae.ctypes.data_as(_dp),
                                                               ctypes.byref(st), ctypes.byref(it))
        return st.value, it.value

    def solve_batch(self, sets, out_name, B, mu0=1.0, maxIter=200, addEye2Hessian=(1e-9, 1e-9)):
        """sets: {entry name: array [B, size]} -> (outputs [B, nOut], status [B], iter [B])"""
        names = list(sets)
        ents = (ctypes.c_int * len(names))(*[self.entry(n) for n in names])
        arrs = [np.ascontiguousarray(sets[n], dtype=np.float64).reshape(B, -1) for n in names]
        for n, a in zip(names, arrs):
            if a.shape[1] != self.size(n):
                raise ValueError("%s expects %d values per instance" % (n, self.size(n)))
        ptrs = (_dp * len(names))(*[a.ctypes.data_as(_dp) for a in arrs])
        nOut = self.size(out_name)
        out = np.empty((B, nOut))
        st = np.zeros(B, dtype=np.int32)
        it = np.zeros(B, dtype=np.int32)
        ae = np.array(addEye2Hessian, dtype=np.float64)
        (self.L.tco_solve_eq_batch if self.eq else self.L.tco_solve_batch)(self.h, len(names), ents, ptrs, self.entry(out_name),
                               out.ctypes.data_as(_dp), st.ctypes.data_as(_i32p),
                               it.ctypes.data_as(_i32p), B, mu0, maxIter, ae.ctypes.data_as(_dp))
        return out, st, it

    def counters(self):
        return dict(flops=int(self.L.tco_flops(self.h)), group_evals=int(self.L.tco_group_evals(self.h)),
                    minF_trials=int(self.L.tco_minF_trials(self.h)))

    def reset_counters(self):
        self.L.tco_reset_counters(self.h)

    def __del__(self):
        try:
            self.L.tco_free(self.h)
        except Exception:
            pass


class ReferenceLoopOracle(Oracle):
    """Same tape interpreter, but solve() runs the reference's own ipmPD_CSsolver.c
    (oracle/_ref build).  Only available where oracle/_ref/*.so exists."""

    def __init__(self, tape_path, tag, defines, solver="ipmPD_CSsolver"):
        lib = _build.build_ref(tag, defines, solver=solver)
        if lib is None:
            raise FileNotFoundError("oracle/_ref build for %s not available" % tag)
        super().__init__(tape_path, lib=lib)
        self.L.tcref_bind.argtypes = [ctypes.c_void_p]
        self._loop = getattr(self.L, solver)           # the reference's own symbol
        self._loop.argtypes = [_dp, _i32p, _i32p, _dp, _i32p, _i32p, _dp]

    def solve(self, mu0=1.0, maxIter=200, saveIter=-1, addEye2Hessian=(1e-9, 1e-9)):
        self.L.tcref_bind(self.h)
        mu = ctypes.c_double(mu0)
        mi, si = ctypes.c_int32(maxIter), ctypes.c_int32(saveIter)
        ae = np.array(addEye2Hessian, dtype=np.float64)
        st, it, tm = ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_double(0)
        self._loop(ctypes.byref(mu), ctypes.byref(mi), ctypes.byref(si), ae.ctypes.data_as(_dp), ctypes.byref(st),
                   ctypes.byref(it), ctypes.byref(tm))
        return st.value, it.value


class DeviceLoopOracle(Oracle):
    """Same tape interpreter, but solve() runs the DEVICE Newton loop
    (tenscalc_amd/csrc/ipm_newton.hip.inc) compiled for the host by oracle/build.py:build_devloop."""

    def __init__(self, tape_path, tag, defines, solver="ipmPD_CSsolver"):
        lib = _build.build_devloop(tag, defines, solver=solver)
        super().__init__(tape_path, lib=lib)
        self.L.tcdev_solve.argtypes = [ctypes.c_void_p, ctypes.c_double, ctypes.c_int, ctypes.c_double,
                                       ctypes.c_double, _i32p, _i32p]

    def solve(self, mu0=1.0, maxIter=200, saveIter=-1, addEye2Hessian=(1e-9, 1e-9)):
        st, it = ctypes.c_int32(0), ctypes.c_int32(0)
        self.L.tcdev_solve(self.h, float(mu0), int(maxIter), float(addEye2Hessian[0]),
                           float(addEye2Hessian[1]), ctypes.byref(st), ctypes.byref(it))
        return st.value, it.value

    def trace(self):
        """names of the entry points the last solve() called, in order"""
        self.L.tcdev_trace.argtypes = [_i32p, ctypes.c_int]
        self.L.tco_entry_name.restype = ctypes.c_char_p
        self.L.tco_entry_name.argtypes = [ctypes.c_void_p, ctypes.c_int]
        buf = np.zeros(1 << 20, dtype=np.int32)
        n = self.L.tcdev_trace(buf.ctypes.data_as(_i32p), buf.size)
        return [self.L.tco_entry_name(self.h, int(e)).decode() for e in buf[:min(n, buf.size)]]
