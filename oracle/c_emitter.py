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


FLAT_ABOVE = 20000      # tapes beyond this many instructions: every value in m[], group bodies cut into small
CHUNK = 256             # functions spread over translation units compiled side by side (gcc -O1 is superlinear in
                        # the size of a function: l1l2's 54 k instructions as one function per group take an hour)


def _prelude(L, T, storage):
    return ("/* GENERATED CPU twin (oracle/c_emitter.py) -- test infrastructure */\n"
            "#include <math.h>\n#include <float.h>\n#include <stdint.h>\n#include <string.h>\n"
            f"{storage} double m[{max(1, L.nPriv)}];\n{storage} double S[{max(1, L.nExt)}];\n"
            f"{storage} unsigned char groupStatus[{T.nGroups + 1}];\n"
            "#define MM(k) (m[k])\n#define SX(k) (S[k])\n#define TC_DIV(x, d) ((x) / (d))\n#define TC_NUM(x, d) (x)\n#define TC_POW_NEG(x, y) pow((x), (y))\n#define TC_EXP(x) exp(x)\n#define TC_LOG(x) log(x)\n#define TC_SIN(x) sin(x)\n#define TC_COS(x) cos(x)\n#define TC_ATAN(x) atan(x)\n"
            "static inline double tc_min(double a, double x) { return (a > x) ? x : a; }\n"
            "static inline double tc_max(double a, double x) { return (a < x) ? x : a; }\n"
            "static inline double tc_clp(double a, double x, double dx) { if (dx < 0) { const double t = -x / dx; if (a > t) a = t; } return a; }\n")


def _chunks(body):
    """cut a group body (flat statements, possibly brace blocks) into pieces of about CHUNK lines at brace depth 0"""
    out, cur, depth = [], [], 0
    for ln in body.splitlines(keepends=True):
        cur.append(ln)
        depth += ln.count("{") - ln.count("}")
        if depth == 0 and len(cur) >= CHUNK:
            out.append("".join(cur))
            cur = []
    if cur:
        out.append("".join(cur))
    return out


def emit_c(spec, flat=None, ntu=16):
    """-> (main source, input names, input sizes, nOut[, extra translation units when flat])"""
    T = spec.tape
    if flat is None:
        flat = T.n > FLAT_ABOVE
    if flat:
        L = Lowering(T, None, roll=False, lanes=1, spill_all=True)
        P = Printer(L, roll=False)
    else:
        L = Lowering(T, getattr(spec, "code", None), roll=True, lanes=1)
        P = Printer(L, roll=True)
    cls = spec.classname
    extra = []
    if flat:
        w = [_prelude(L, T, "")]
        tus = [[_prelude(L, T, "extern")] for _ in range(ntu)]
        sizes = [0] * ntu
        for g in range(1, T.nGroups + 1):
            parts = _chunks(P.group_body(g))
            for k, part in enumerate(parts):
                j = sizes.index(min(sizes))
                tus[j].append(f"void group{g}_{k}(void) {{\n{part}}}\n")
                sizes[j] += len(part)
                w.append(f"void group{g}_{k}(void);\n")
            w.append(f"static void group{g}(void) {{\n" + "".join(f"  group{g}_{k}();\n" for k in range(len(parts))) + "}\n")
        extra = ["".join(t) for t in tus if len(t) > 1]
        return _emit_entry_points(spec, T, L, P, cls, w) + (extra,)
    w = [_prelude(L, T, "static")]
    for g in range(1, T.nGroups + 1):
        w.append(f"static void group{g}(void) {{\n{P.group_body(g)}}}\n")
    return _emit_entry_points(spec, T, L, P, cls, w) + ([],)


def _emit_entry_points(spec, T, L, P, cls, w):
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
    src, names, sizes, nout, extra = emit_c(spec)
    if not os.path.exists(REF_SOLVER):
        return (out, names, sizes, nout) if os.path.exists(out) else None
    os.makedirs(os.path.join(HERE, "_ref"), exist_ok=True)
    cfile = os.path.join(HERE, "_ref", f"{cls}_cpu.c")
    stamp = cfile + ".stamp"
    import hashlib
    digest = hashlib.sha256((src + "".join(extra) + opt).encode()).hexdigest()
    if os.path.exists(out) and os.path.exists(stamp) and open(stamp).read() == digest:
        return out, names, sizes, nout
    with open(cfile, "w") as f:
        f.write(src)

    def fmt(v):
        return "%e" % v if isinstance(v, float) else str(int(v))
    dflags = ["-D%s=%s" % (k, fmt(spec.defines[k])) for k in _DEFINE_NAMES]
    dflags += ["-DverboseLevel=0", "-DEXPORT=", '-DsaveNamePrefix="x"']
    obj = cfile[:-2] + "_loop.o"
    cc = ["gcc", "-w", opt, "-ffp-contract=off", "-std=gnu99", "-fPIC"]
    cmds = [cc + ["-c"] + dflags + [REF_SOLVER, "-o", obj]]
    objs = [obj]
    for k, tu in enumerate(extra):          # big tapes: the group bodies as translation units compiled side by side
        tf = cfile[:-2] + f"_tu{k}.c"
        with open(tf, "w") as f:
            f.write(tu)
        cmds.append(cc + ["-c", tf, "-o", tf[:-2] + ".o"])
        objs.append(tf[:-2] + ".o")
    jobs = max(1, min(len(cmds), int(os.environ.get("TC_BUILD_JOBS", "0")) or (os.cpu_count() or 1)))
    running = []
    for cmd in cmds + [None] * jobs:
        while running and (cmd is None or len(running) >= jobs):
            c_, p_ = running.pop(0)
            _, err = p_.communicate()
            if p_.returncode:
                raise RuntimeError("cpu twin build failed: %s\n%s" % (" ".join(c_), err[-3000:]))
        if cmd is not None:
            running.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)))
    r = subprocess.run(cc + ["-shared", cfile] + objs + ["-o", out, "-lm"], capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError("cpu twin link failed:\n%s" % r.stderr[-3000:])
    for o_ in objs[1:]:
        os.remove(o_)
        os.remove(o_[:-2] + ".c")
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

    @classmethod
    def from_parts(cls, path, names, sizes, nout):
        """a twin from what a worker process needs of it (bench.py's all-cores baseline: no tape is rebuilt there)"""
        import ctypes
        self = cls.__new__(cls)
        self.path, self.names, self.sizes, self.nout = path, list(names), list(sizes), int(nout)
        self.L = ctypes.CDLL(path)
        self._ct = ctypes
        return self

    def parts(self):
        return self.path, list(self.names), list(self.sizes), int(self.nout)

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
