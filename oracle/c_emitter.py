"""ORACLE - test infrastructure only.

CPU twin of a generated solver class in the reference's own style: the scalar tape printed as
single-instance C on one static scratchbook with lazily evaluated dependency groups and
set/get/copy functions under the reference's names (what @csparse/writeCfunctionpergroup.m +
writeCinstructionsC emit), compiled with gcc -O1 and linked against the reference's Newton loop
lib/ipmPD_CSsolver.c taken from /root/reference where it lies (never copied).  The statement
printer is shared with the HIP emitter (lanes=1 lowering), so this is the same tape in the same
evaluation order.  Used as the `cpu_baseline` ("port": the tape text comes from our emitter, not
from the reference's MATLAB generator) and as one more cross-check of the interpreter oracle.

Output: oracle/_ref/lib<cls>_cpu.so (git-ignored; travels to the GPU box prebuilt).
"""
import os
import subprocess

import numpy as np

from tenscalc_amd.hip_emitter import Printer
from tenscalc_amd.lowering import Lowering

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SOLVER = "/root/reference/lib/ipmPD_CSsolver.c"

_DEFINE_NAMES = ["nU", "nG", "nF", "gradTolerance", "addEye2HessianUtolerance", "equalTolerance",
                 "desiredDualityGap", "scaleCost", "scaleInequalities", "scaleEqualities", "alphaMin",
                 "alphaMax", "coupledAlphas", "setAddEye2Hessian", "adjustAddEye2Hessian", "useInertia",
                 "muFactorAggressive", "muFactorConservative", "delta", "skipAffine",
                 "smallerNewtonMatrix", "useLDL", "useUmfpack", "allowSave", "debugConvergence", "profiling"]


def emit_c(spec):
    T = spec.tape
    L = Lowering(T, getattr(spec, "code", None), roll=True, lanes=1)
    P = Printer(L, roll=True)
    cls = spec.classname
    w = ["/* GENERATED CPU twin (oracle/c_emitter.py) -- test infrastructure */\n"
         "#include <math.h>\n#include <float.h>\n#include <stdint.h>\n#include <string.h>\n"
         f"static double m[{max(1, L.nPriv)}];\nstatic double S[{max(1, L.nExt)}];\n"
         f"static unsigned char groupStatus[{T.nGroups + 1}];\n"
         "#define MM(k) (m[k])\n#define SX(k) (S[k])\n#define TC_DIV(x, d) ((x) / (d))\n#define TC_NUM(x, d) (x)\n#define TC_POW_NEG(x, y) pow((x), (y))\n"
         "static inline double tc_min(double a, double x) { return (a > x) ? x : a; }\n"
         "static inline double tc_max(double a, double x) { return (a < x) ? x : a; }\n"
         "static inline double tc_clp(double a, double x, double dx) { if (dx < 0) { const double t = -x / dx; if (a > t) a = t; } return a; }\n"]
    for g in range(1, T.nGroups + 1):
        w.append(f"static void group{g}(void) {{\n{P.group_body(g)}}}\n")
    w.append("#define NEED(g) do { if (!groupStatus[g]) { groupStatus[g] = 1; group##g(); } } while (0)\n")

    def inval(ch):
        return "".join(f"  groupStatus[{g}] = 0;\n" for g in ch)
    for e in T.sets:
        w.append(f"void {e['name']}(const double *input) {{\n")
        for k, i in enumerate(e["ins"]):
            if e["external"]:
                w.append(f"  S[{L.ext_index[i]}] = input[{k}];\n")
            elif L.pslot[i] >= 0:
                w.append(f"  m[{L.pslot[i]}] = input[{k}];\n")
        w.append(inval(e["children"]) + "}\n")
    for e in T.gets:
        args = ", ".join(f"double *y{k + 1}" for k in range(len(e["outputs"])))
        w.append(f"void {e['name']}({args}) {{\n" + "".join(f"  NEED({g});\n" for g in e["parents"]))
        for k, o in enumerate(e["outputs"]):
            for j, i in enumerate(o["ins"]):
                w.append(f"  y{k + 1}[{j}] = {P.ref(i)};\n")
        w.append("}\n")
    for e in T.copies:
        w.append(f"void {e['name']}(void) {{\n" + "".join(f"  NEED({g});\n" for g in e["parents"]))
        for d, s in zip(e["dst"], e["src"]):
            if L.pslot[d] >= 0:
                w.append(f"  m[{L.pslot[d]}] = {P.ref(s)};\n")
        w.append(inval(e["children"]) + "}\n")
    # batch driver: instance k reads in[s][k*size_s..], solves with the reference loop, writes outputs
    ins = [e for e in T.sets if e["external"]]
    outs = T.entry(f"{cls}_getOutputs")[2]["outputs"]
    w.append("extern void ipmPD_CSsolver(double*, int32_t*, int32_t*, double*, int32_t*, int32_t*, double*);\n")
    w.append("void tc_cpu_solve_batch(const double *const *in, double *out, int32_t *status, int32_t *iter, long B,\n"
             "                        double mu0, int32_t maxIter, const double *addEye) {\n"
             "  double ae[2] = {addEye[0], addEye[1]}, tm = 0;\n  int32_t saveIter = -1;\n"
             f"  const long nOut = {sum(len(o['ins']) for o in outs)};\n"
             "  for (long k = 0; k < B; k++) {\n")
    for s_, e in enumerate(ins):
        w.append(f"    {e['name']}(in[{s_}] + k * {len(e['ins'])});\n")
    w.append("    double mu = mu0; int32_t mi = maxIter; status[k] = 0;\n"
             "    ipmPD_CSsolver(&mu, &mi, &saveIter, ae, &status[k], &iter[k], &tm);\n"
             f"    {cls}_getOutputs(" + ", ".join(f"out + k * nOut + {off}" for off in
                                                   np.cumsum([0] + [len(o["ins"]) for o in outs])[:-1]) + ");\n"
             "  }\n}\n")
    return "".join(w), [e["varname"] for e in ins], [len(e["ins"]) for e in ins], sum(len(o["ins"]) for o in outs)


def build_cpu_twin(spec, opt="-O1"):
    """-> (path of lib<cls>_cpu.so, input names, input sizes, nOut) or None if it cannot be built
    here and no prebuilt library exists"""
    cls = spec.classname
    out = os.path.join(HERE, "_ref", f"lib{cls}_cpu.so")
    src, names, sizes, nout = emit_c(spec)
    if not os.path.exists(REF_SOLVER):
        return (out, names, sizes, nout) if os.path.exists(out) else None
    os.makedirs(os.path.join(HERE, "_ref"), exist_ok=True)
    cfile = os.path.join(HERE, "_ref", f"{cls}_cpu.c")
    stamp = cfile + ".stamp"
    import hashlib
    digest = hashlib.sha256((src + opt).encode()).hexdigest()
    if os.path.exists(out) and os.path.exists(stamp) and open(stamp).read() == digest:
        return out, names, sizes, nout
    with open(cfile, "w") as f:
        f.write(src)

    def fmt(v):
        return "%e" % v if isinstance(v, float) else str(int(v))
    dflags = ["-D%s=%s" % (k, fmt(spec.defines[k])) for k in _DEFINE_NAMES]
    dflags += ["-DverboseLevel=0", "-DEXPORT=", '-DsaveNamePrefix="x"']
    obj = cfile[:-2] + "_loop.o"
    for cmd in (["gcc", "-w", opt, "-ffp-contract=off", "-std=gnu99", "-fPIC", "-c"] + dflags + [REF_SOLVER, "-o", obj],
                ["gcc", "-w", opt, "-ffp-contract=off", "-std=gnu99", "-fPIC", "-shared", cfile, obj, "-o", out, "-lm"]):
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError("cpu twin build failed: %s\n%s" % (" ".join(cmd), r.stderr[-3000:]))
    os.remove(obj)
    with open(stamp, "w") as f:
        f.write(digest)
    return out, names, sizes, nout


class CpuTwin:
    def __init__(self, spec):
        import ctypes
        r = build_cpu_twin(spec)
        if r is None:
            raise FileNotFoundError("CPU twin of %s not available" % spec.classname)
        self.path, self.names, self.sizes, self.nout = r
        self.L = ctypes.CDLL(self.path)
        self._ct = ctypes

    def solve_batch(self, inputs, B, mu0=1.0, maxIter=200, addEye2Hessian=(1e-9, 1e-9)):
        """inputs: {name: [B, size]} -> (out [B, nOut], status, iter)"""
        ct = self._ct
        dp = ct.POINTER(ct.c_double)
        arrs = [np.ascontiguousarray(inputs[n], dtype=np.float64).reshape(B, s) for n, s in zip(self.names, self.sizes)]
        ptrs = (dp * len(arrs))(*[a.ctypes.data_as(dp) for a in arrs])
        out = np.empty((B, self.nout))
        st = np.zeros(B, dtype=np.int32)
        it = np.zeros(B, dtype=np.int32)
        ae = np.array(addEye2Hessian, dtype=np.float64)
        self.L.tc_cpu_solve_batch.argtypes = [ct.POINTER(dp), dp, ct.POINTER(ct.c_int32), ct.POINTER(ct.c_int32),
                                              ct.c_long, ct.c_double, ct.c_int32, dp]
        self.L.tc_cpu_solve_batch(ptrs, out.ctypes.data_as(dp), st.ctypes.data_as(ct.POINTER(ct.c_int32)),
                                  it.ctypes.data_as(ct.POINTER(ct.c_int32)), B, float(mu0), int(maxIter),
                                  ae.ctypes.data_as(dp))
        return out, st, it
