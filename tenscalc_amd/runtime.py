"""Host-side mirror of the reference's generated MATLAB class for one solver:
``obj=<cls>(); setP_<param>(obj,v); setV_<var>(obj,v); [status,iter,time]=solve(obj,mu0,
int32(maxIter),int32(saveIter)[,addEye2Hessian]); [y1,...]=getOutputs(obj)``
(/root/reference/README.md:203-215, lib/cmex2optimizeCS.m:153-195,280-294,338-354), reaching the
HIP class library through its C-ABI (include/tenscalc_hip.h) with ctypes -- the stand-in, in this
MATLAB-less image, for the mex gateway described in INTEGRATION.md.

Batched extension: every set takes ``[size..., B]`` (MATLAB order: batch is the
last = slowest dimension; a C-contiguous numpy array of shape ``[B, *reversed(size)]`` is the
same memory), ``solve`` returns ``status, iter`` as int32 ``[B]``, ``getOutputs`` returns arrays
with a trailing batch dimension.  There is no CPU fallback: a missing library or a failing HIP
call raises.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

_dp = ctypes.POINTER(ctypes.c_double)
_i32p = ctypes.POINTER(ctypes.c_int32)

LAYOUT_INSTANCE_MAJOR = 0
LAYOUT_BATCH_MINOR = 1


class TensCalcHipError(RuntimeError):
    pass


def _bind(path):
    # PyTorch-ROCm ships its own HIP runtime; load it first so the class library (linked against
    # libamdhip64) binds to the same runtime instead of bringing a second one into the process.
    try:
        import torch  # noqa: F401
    except ImportError:  # pragma: no cover - torch is plumbing, not a requirement of the C-ABI
        pass
    if not os.path.exists(path):
        raise TensCalcHipError(f"solver class library {path} not found: build it first "
                               "(SolverSpec.build() / __graft_entry__.build())")
    L = ctypes.CDLL(path, mode=ctypes.RTLD_LOCAL)
    c = ctypes
    L.tc_classname.restype = c.c_char_p
    L.tc_last_error.restype = c.c_char_p
    L.tc_input_name.restype = c.c_char_p
    L.tc_input_name.argtypes = [c.c_int]
    L.tc_output_name.restype = c.c_char_p
    L.tc_output_name.argtypes = [c.c_int]
    for f in ("tc_input_numel", "tc_output_numel", "tc_output_offset"):
        getattr(L, f).restype = c.c_int64
        getattr(L, f).argtypes = [c.c_int]
    for f in ("tc_input_dim", "tc_output_dim"):
        getattr(L, f).restype = c.c_int64
        getattr(L, f).argtypes = [c.c_int, c.c_int]
    for f in ("tc_input_ndim", "tc_output_ndim", "tc_input_is_variable"):
        getattr(L, f).argtypes = [c.c_int]
    L.tc_define_double.restype = c.c_double
    L.tc_define_double.argtypes = [c.c_char_p]
    L.tc_define_int.restype = c.c_int64
    L.tc_define_int.argtypes = [c.c_char_p]
    L.tc_flops_static.restype = c.c_int64
    L.tc_flops_static.argtypes = [c.c_char_p]
    L.tc_create.argtypes = [c.c_int64, c.c_int, c.POINTER(c.c_void_p)]
    L.tc_destroy.argtypes = [c.c_void_p]
    L.tc_set.argtypes = [c.c_void_p, c.c_char_p, _dp, c.c_int64]
    L.tc_set_device.argtypes = [c.c_void_p, c.c_char_p, c.c_void_p, c.c_int64, c.c_int, c.c_void_p]
    L.tc_set_from_output.argtypes = [c.c_void_p, c.c_char_p, c.c_char_p, c.c_void_p]
    L.tc_solve.argtypes = [c.c_void_p, _dp, _i32p, _i32p, _dp, _i32p, _i32p, _dp]
    L.tc_solve_async.argtypes = [c.c_void_p, c.c_double, c.c_int32, _dp, c.c_void_p]
    L.tc_get_outputs.argtypes = [c.c_void_p, c.POINTER(_dp), c.c_int64]
    L.tc_get_device.argtypes = [c.c_void_p, c.POINTER(c.c_void_p), c.POINTER(c.c_void_p),
                                c.POINTER(c.c_void_p)]
    L.tc_synchronize.argtypes = [c.c_void_p]
    L.tc_get_stream.argtypes = [c.c_void_p, c.POINTER(c.c_void_p)]
    L.tc_probe.argtypes = [c.c_void_p, c.c_char_p, c.c_double, _dp, _dp, c.c_int64]
    L.tc_stats_rows.restype = c.c_int64
    L.tc_get_stats.argtypes = [c.c_void_p, _dp]
    L.tc_profiling_view.argtypes = [c.c_void_p, c.c_char_p]
    L.tc_trace.argtypes = [c.c_void_p, c.c_int64, c.c_char_p]
    L.tc_probe_numel.restype = c.c_int64
    L.tc_probe_numel.argtypes = [c.c_char_p]
    return L


class HipSolver:
    """One loaded solver class with a batch of B instances resident on one GPU."""

    def __init__(self, library, B=1, device=0):
        self._L = _bind(library)
        self.library = library
        self.classname = self._L.tc_classname().decode()
        self.B = int(B)
        self.device = int(device)
        self.inputs = []
        for k in range(self._L.tc_num_inputs()):
            nd = self._L.tc_input_ndim(k)
            self.inputs.append(dict(
                name=self._L.tc_input_name(k).decode(), is_variable=bool(self._L.tc_input_is_variable(k)),
                numel=int(self._L.tc_input_numel(k)), size=tuple(int(self._L.tc_input_dim(k, d)) for d in range(nd))))
        self.outputs = []
        for k in range(self._L.tc_num_outputs()):
            nd = self._L.tc_output_ndim(k)
            self.outputs.append(dict(
                name=self._L.tc_output_name(k).decode(), numel=int(self._L.tc_output_numel(k)),
                offset=int(self._L.tc_output_offset(k)),
                size=tuple(int(self._L.tc_output_dim(k, d)) for d in range(nd))))
        h = ctypes.c_void_p()
        self._check(self._L.tc_create(self.B, self.device, ctypes.byref(h)))
        self._h = h
        for inp in self.inputs:
            pre = "setV_" if inp["is_variable"] else "setP_"
            setattr(self, pre + inp["name"], (lambda v, _n=inp["name"]: self.set(_n, v)))

    # -- helpers -----------------------------------------------------------------------
    def _check(self, rc):
        if rc != 0:
            raise TensCalcHipError(f"{self.classname}: error {rc}: {self._L.tc_last_error().decode()}")

    def define(self, name):
        return self._L.tc_define_double(name.encode())

    def static_count(self, what):
        return int(self._L.tc_flops_static(what.encode()))

    def close(self):
        if getattr(self, "_h", None):
            self._L.tc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- setP_/setV_ ----------------------------------------------------------------------
    def set(self, name, value):
        """value: host array with B*numel doubles, instance-major ([B, ...] C-order ==
        MATLAB [size...,B] column-major), or a torch CUDA tensor (see set_device)."""
        if hasattr(value, "is_cuda") and value.is_cuda:
            return self.set_device(name, value)
        inp = self._input(name)
        v = np.ascontiguousarray(value, dtype=np.float64).reshape(-1)
        if v.size == inp["numel"] and self.B > 1:
            v = np.tile(v, self.B)           # same value for every instance
        if v.size != inp["numel"] * self.B:
            raise ValueError(f"{name}: expected {self.B}x{inp['numel']} values, got {v.size}")
        self._check(self._L.tc_set(self._h, name.encode(), v.ctypes.data_as(_dp), self.B))

    def set_device(self, name, tensor, layout=LAYOUT_INSTANCE_MAJOR, stream=None):
        """tensor: contiguous float64 torch tensor already on this GPU; layout says whether it is
        [B, numel] (instance-major) or [numel, B] (structure-of-arrays)."""
        inp = self._input(name)
        if tensor.dtype.__str__() != "torch.float64" or not tensor.is_contiguous():
            raise ValueError("set_device needs a contiguous float64 tensor")
        if tensor.numel() != inp["numel"] * self.B:
            raise ValueError(f"{name}: expected {self.B}x{inp['numel']} values")
        self._check(self._L.tc_set_device(self._h, name.encode(), ctypes.c_void_p(tensor.data_ptr()),
                                          self.B, int(layout), ctypes.c_void_p(stream or 0)))

    def set_from_output(self, input_name, output_name, stream=None):
        """input <- output of the previous solve, device to device (warm start, receding horizon)"""
        self._check(self._L.tc_set_from_output(self._h, input_name.encode(), output_name.encode(),
                                               ctypes.c_void_p(stream or 0)))

    def _input(self, name):
        for inp in self.inputs:
            if inp["name"] == name:
                return inp
        raise KeyError(f"class {self.classname} has no parameter/variable '{name}'")

    # -- solve ---------------------------------------------------------------------------------
    def solve(self, mu0=1.0, maxIter=200, saveIter=-1, addEye2Hessian=(1e-9, 1e-9)):
        """-> (status[B] int32, iter[B] int32, time seconds)"""
        mu = ctypes.c_double(float(mu0))
        mi, si = ctypes.c_int32(int(maxIter)), ctypes.c_int32(int(saveIter))
        ae = np.array(addEye2Hessian, dtype=np.float64).reshape(2)
        st = np.zeros(self.B, dtype=np.int32)
        it = np.zeros(self.B, dtype=np.int32)
        tm = ctypes.c_double(0)
        self._check(self._L.tc_solve(self._h, ctypes.byref(mu), ctypes.byref(mi), ctypes.byref(si),
                                     ae.ctypes.data_as(_dp), st.ctypes.data_as(_i32p),
                                     it.ctypes.data_as(_i32p), ctypes.byref(tm)))
        return st, it, tm.value

    def solve_async(self, mu0=1.0, maxIter=200, addEye2Hessian=(1e-9, 1e-9), stream=None):
        ae = np.array(addEye2Hessian, dtype=np.float64).reshape(2)
        self._check(self._L.tc_solve_async(self._h, float(mu0), int(maxIter), ae.ctypes.data_as(_dp),
                                           ctypes.c_void_p(stream or 0)))

    def stream(self):
        """the handle's own stream (hipStream_t as an int): wrap it with torch.cuda.ExternalStream to enqueue there"""
        p = ctypes.c_void_p(0)
        self._check(self._L.tc_get_stream(self._h, ctypes.byref(p)))
        return int(p.value or 0)

    def synchronize(self):
        self._check(self._L.tc_synchronize(self._h))

    # -- getOutputs ---------------------------------------------------------------------------------
    def getOutputs(self, as_dict=False):
        """arrays of shape [B, *reversed(size)] (== MATLAB [size..., B])"""
        bufs = [np.empty((self.B, o["numel"])) for o in self.outputs]
        ptrs = (_dp * len(bufs))(*[b.ctypes.data_as(_dp) for b in bufs])
        self._check(self._L.tc_get_outputs(self._h, ptrs, self.B))
        outs = [b.reshape((self.B,) + tuple(reversed(o["size"]))) for b, o in zip(bufs, self.outputs)]
        if as_dict:
            return {o["name"]: a for o, a in zip(self.outputs, outs)}
        return outs

    def device_views(self):
        """(outputs_ptr [TC_NOUT][B] doubles, status_ptr, iter_ptr) raw device addresses"""
        o, s, i = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        self._check(self._L.tc_get_device(self._h, ctypes.byref(o), ctypes.byref(s), ctypes.byref(i)))
        return o.value, s.value, i.value

    def stats(self):
        """diagnostic (stats=True) builds: per-group cycles/calls -> array [rows, B]"""
        rows = int(self._L.tc_stats_rows())
        out = np.zeros((rows, self.B))
        self._check(self._L.tc_get_stats(self._h, out.ctypes.data_as(_dp)))
        return out

    def profilingView(self, filename):
        """the reference class's profilingView(obj, filename) (writeCprofiling.c:8-83): per dependency group executions and
        time of the last solve, averaged over the batch; classes built with stats=True only"""
        self._check(self._L.tc_profiling_view(self._h, os.fsencode(filename)))

    def trace(self, instance, filename):
        """the reference's solverVerboseLevel >= 3 iteration table (ipmPD_CSsolver.c:248-345,775-863) of ONE instance of the
        last batched solve, written to `filename`; classes generated with solverVerboseLevel >= 3 only"""
        self._check(self._L.tc_trace(self._h, int(instance), os.fsencode(filename)))

    def probe(self, get_name, mu0=1.0, addEye2Hessian=(1e-9, 1e-9)):
        """evaluate a tape get entry point after the solver's initialisation sequence -> [B, numel]"""
        n = int(self._L.tc_probe_numel(get_name.encode()))
        if n < 0:
            raise KeyError(get_name)
        out = np.empty((self.B, n))
        ae = np.array(addEye2Hessian, dtype=np.float64).reshape(2)
        self._check(self._L.tc_probe(self._h, get_name.encode(), float(mu0), ae.ctypes.data_as(_dp),
                                     out.ctypes.data_as(_dp), self.B))
        return out


    def saveWW(self, filename=None, instance=0, mu0=1.0, addEye2Hessian=(1e-9, 1e-9)):
        """``saveWW__`` of classes generated with allowSave=True: the values of the Newton matrix of `instance`, evaluated
        right after the initialisation sequence (the first iteration's WW; the reference writes the one of iteration
        saveIter, ipmPD_CSsolver.c:561-566), written as ``<cls>_WW.values`` next to the ``<cls>_WW.subscripts`` file the code
        generator left beside the library (tenscalc_amd/typical.py).  A later code generation in the same folder picks
        the pair up and orders the symbolic LU with these values instead of random ones (sparsity_lu.m:36-57)."""
        import os
        from . import typical
        base = os.path.join(os.path.dirname(os.path.abspath(self.library)), self.classname + "_WW")
        if not os.path.exists(base + ".subscripts"):
            raise TensCalcHipError("saveWW: %s.subscripts not found (generate the class with allowSave=True)" % base)
        magic, osize, subs = typical.read_subscripts(base + ".subscripts")
        nd = len(osize)
        dense = self.probe("saveWW__", mu0=mu0, addEye2Hessian=addEye2Hessian)[int(instance)]
        dense = dense.reshape(osize, order="F")
        vals = dense[tuple(subs[:, k] for k in range(nd))]
        return typical.write_values(filename or (base + ".values"), vals, magic)


def load_class(spec, B=1, device=0, **build_kw):
    """``obj=feval(classname)``: build (if needed) and instantiate."""
    so = spec.build(**build_kw)
    return HipSolver(so, B=B, device=device)
