"""Tape -> HIP lowering (the step the build retargets: SURVEY.md 8a row A9).

The reference prints the scalar tape as straight-line C on one global scratchbook
(/root/reference/lib/@csparse/writeCfunctionpergroup.m:386-504 + ``writeCinstructionsC``,
lib/csparse/instructionsTableUTHash.c:790-1609).  This emitter prints the *same tape* as
CDNA4 HIP where **one thread owns one problem instance**:

* parameters and initial values live in HBM as structure-of-arrays ``S[element][B]`` so that
  a wavefront's 64 instances read 64 consecutive doubles (the batch is the coalesced axis);
* values that cross dependency-group boundaries live in a per-thread private array ``m[]``;
* values used only inside their group are SSA temporaries (registers);
* ``I_load`` constants are literals;
* each dependency group is one ``__device__`` function, evaluated lazily through a per-thread
  bitmask, exactly the ``groupStatus[]`` protocol of the reference
  (writeCfunctionpergroup.m:44,653-657);
* the Newton loop (csrc/ipm_newton.hip.inc) runs on the device, so a whole solve is one launch.

Per-opcode arithmetic (operand order, ``m[r]+=`` chunking of long sums) follows
instructionsTableUTHash.c:950-1599 so that with ``fp_contract='off'`` results are bit-identical
to the CPU oracle; ``fp_contract='fast'`` lets the compiler form FMAs.
"""
from __future__ import annotations

import hashlib
import os
import subprocess

import numpy as np

from . import instructions as I

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
MAX_TERMS_PERLINE = 100  # instructionsTableUTHash.c:173

_UNARY = {
    I.I_exp: "exp({0})", I.I_log: "log({0})", I.I_cos: "cos({0})", I.I_minus_cos: "-cos({0})",
    I.I_sin: "sin({0})", I.I_minus_sin: "-sin({0})", I.I_round: "round({0})",
    I.I_ceil: "ceil({0})", I.I_floor: "floor({0})", I.I_abs: "fabs({0})",
    I.I_sign: "(({0}>=0)?1.0:-1.0)", I.I_sqrt: "sqrt({0})", I.I_Dsqrt: ".5*pow({0},-.5)",
    I.I_DDsqrt: "-.25*pow({0},-1.5)", I.I_sqr: "{0}*{0}", I.I_2times: "2*{0}", I.I_2: "2.0",
    I.I_cube: "{0}*{0}*{0}", I.I_3sqr: "3*{0}*{0}", I.I_6times: "6*{0}", I.I_atan: "atan({0})",
    I.I_Datan: "1/(1+{0}*{0})", I.I_DDatan: "-2*{0}/(1+{0}*{0})/(1+{0}*{0})",
    I.I_srelu: "log(1+exp({0}))", I.I_dsrelu: "1/(1+exp(-{0}))",
    I.I_ddsrelu: "1/(2+exp(-{0})+exp({0}))", I.I_relu: "(({0}>0)?{0}:0.0)",
    I.I_heaviside: "(({0}>0)?1.0:0.0)", I.I_inv: "1/{0}", I.I_minus_inv_sqr: "-1/({0}*{0})",
    I.I_2_inv_cube: "2/({0}*{0}*{0})",
}


def _lit(v):
    v = float(v)
    if v != v or v in (float("inf"), float("-inf")):
        raise ValueError("non-finite constant in tape")
    s = repr(v)
    if "e" not in s and "." not in s:
        s += ".0"
    return "(" + s + ")" if v < 0 else s


class Lowering:
    """Storage classification + statement printer for one tape."""

    def __init__(self, tape):
        T = self.T = tape
        n = T.n
        types, opStart, ops = T.types, T.opStart, T.ops
        # who reads what -------------------------------------------------------------
        self.external = np.zeros(n + 1, dtype=bool)
        self.ext_index = np.full(n + 1, -1, dtype=np.int64)
        self.inputs = []            # external sets in declaration order
        off = 0
        for e in T.sets:
            if e["external"]:
                self.external[e["ins"]] = True
                self.ext_index[e["ins"]] = off + np.arange(len(e["ins"]))
                self.inputs.append(dict(name=e["varname"], entry=e["name"], size=e["size"],
                                        numel=len(e["ins"]), off=off))
                off += len(e["ins"])
        self.nExt = off
        out_needed = np.zeros(n + 1, dtype=bool)   # read by a get / copy source
        for e in T.gets:
            for o in e["outputs"]:
                out_needed[o["ins"]] = True
        for e in T.copies:
            out_needed[e["src"]] = True
        # group-local: every reader is in the same group, and no entry point reads it
        local = np.zeros(n + 1, dtype=bool)
        grp = T.group
        readers_other = np.zeros(n + 1, dtype=bool)
        opl, ost = ops.tolist(), opStart.tolist()
        for i in range(1, n + 1):
            gi = grp[i]
            if gi == 0:
                continue
            for j in opl[ost[i - 1]:ost[i]]:
                if grp[j] != gi:
                    readers_other[j] = True
        for i in range(1, n + 1):
            if grp[i] and not readers_other[i] and not out_needed[i]:
                local[i] = True
        self.local = local
        # private slots: internal sets that someone reads + non-local computed values
        read_any = np.zeros(n + 1, dtype=bool)
        for i in range(1, n + 1):
            if T.used[i]:
                for j in opl[ost[i - 1]:ost[i]]:
                    read_any[j] = True
        read_any |= out_needed
        pslot = np.full(n + 1, -1, dtype=np.int64)
        k = 0
        for i in range(1, n + 1):
            if T.isset[i - 1]:
                if not self.external[i] and read_any[i]:
                    pslot[i] = k
                    k += 1
            elif grp[i] and not local[i]:
                pslot[i] = k
                k += 1
        self.pslot = pslot
        self.nPriv = k

    # -- operand / destination text ---------------------------------------------------
    def ref(self, i):
        T = self.T
        i = int(i)
        if T.isload[i - 1]:
            return _lit(T.pars[T.parStart[i - 1]])
        if self.external[i]:
            return f"SX({self.ext_index[i]})"
        if self.local[i]:
            return f"t{i}"
        if self.pslot[i] < 0:
            raise AssertionError(f"instruction {i} has no storage")
        return f"MM({self.pslot[i]})"

    def assign(self, i, expr):
        if self.local[i]:
            return f"  const double t{i} = {expr};\n"
        return f"  MM({self.pslot[i]}) = {expr};\n"

    def statement(self, i):
        """HIP text computing instruction i (semantics: instructionsTableUTHash.c:950-1599)."""
        typ, par, ops = self.T.instr(i)
        r = [self.ref(o) for o in ops]
        if typ == I.I_sum:
            s = ("+" if par[0] > 0 else "-") + r[0]
            for p, x in zip(par[1:], r[1:]):
                s += ("+" if p > 0 else "-") + x
            return self.assign(i, s)
        if typ == I.I_sumprod:
            nf, ns = int(par[0]), int(par[1])
            terms = ["*".join(r[k * nf:(k + 1) * nf]) for k in range(ns)]
            lines, cur, nSums = [], [], 0
            for k, t in enumerate(terms):
                cur.append(t)
                if k + 1 < ns:
                    nSums += nf
                    if nSums >= MAX_TERMS_PERLINE:
                        lines.append(cur)
                        cur, nSums = [], 0
            lines.append(cur)
            expr = "+".join(lines[0])
            for ln in lines[1:]:
                expr = f"({expr})+({'+'.join(ln)})"
            return self.assign(i, expr)
        if typ == I.I_div:
            return self.assign(i, f"{r[0]}/{r[1]}")
        if typ in (I.I_minus_dot, I.I_minus_dot_div, I.I_plus_minus_dot, I.I_plus_minus_dot_div):
            skip = {I.I_minus_dot: 0, I.I_minus_dot_div: 1, I.I_plus_minus_dot: 1,
                    I.I_plus_minus_dot_div: 2}[typ]
            d = r[skip:]
            dot = "+".join(f"{d[k]}*{d[k + 1]}" for k in range(0, len(d), 2))
            if typ == I.I_minus_dot:
                return self.assign(i, f"-({dot})")
            if typ == I.I_minus_dot_div:
                return self.assign(i, f"-({dot})/{r[0]}")
            if typ == I.I_plus_minus_dot:
                return self.assign(i, f"{r[0]}-({dot})")
            return self.assign(i, f"({r[0]}-({dot}))/{r[1]}")
        if typ == I.I_plus_sqr:
            return self.assign(i, "+".join(f"{x}*{x}" for x in r))
        if typ == I.I_plus_abs:
            return self.assign(i, "+".join(f"fabs({x})" for x in r))
        if typ in (I.I_min, I.I_min0, I.I_max, I.I_max0, I.I_max_abs, I.I_clp):
            b = "  double a_;\n" if False else ""
            if typ == I.I_clp:
                body = "  { double a_=DBL_MAX;\n"
                for k in range(0, len(r), 2):
                    body += f"    if ({r[k + 1]}<0) {{ const double x_=-{r[k]}/{r[k + 1]}; if (a_>x_) a_=x_; }}\n"
            elif typ == I.I_max_abs:
                body = f"  {{ double a_=fabs({r[0]});\n"
                for x in r[1:]:
                    body += f"    if (a_<fabs({x})) a_=fabs({x});\n"
            else:
                cmp_ = ">" if typ in (I.I_min, I.I_min0) else "<"
                if typ in (I.I_min, I.I_max):
                    body = f"  {{ double a_={r[0]};\n"
                    rest = r[1:]
                else:
                    body = "  { double a_=0;\n"
                    rest = r
                for x in rest:
                    body += f"    if (a_{cmp_}{x}) a_={x};\n"
            if self.local[i]:
                return f"  double t{i};\n" + body + f"    t{i}=a_; }}\n"
            return body + f"    MM({self.pslot[i]})=a_; }}\n"
        if typ in _UNARY:
            return self.assign(i, _UNARY[typ].format(r[0] if r else ""))
        raise NotImplementedError(f"HIP emitter: opcode {I.NAMES.get(typ, typ)} not supported")


def _mask(groups):
    m = 0
    for g in groups:
        m |= 1 << (g - 1)
    return m


def emit_class(spec, fp_contract="fast", block=64, stats=False):
    """Return the HIP source text of the solver class."""
    T = spec.tape
    L = Lowering(T)
    cls = spec.classname
    nW = max(1, (T.nGroups + 63) // 64)
    w = []
    w.append(f"// GENERATED by tenscalc_amd.hip_emitter for class '{cls}' -- do not edit.\n"
             f"// tape: {T.n} instructions, {T.nGroups} dependency groups, {L.nPriv} private doubles,"
             f" {L.nExt} input doubles per instance\n")
    w.append("#include <hip/hip_runtime.h>\n#include <math.h>\n#include <float.h>\n#include <stdint.h>\n\n")
    w.append(f'#define TC_CLASSNAME "{cls}"\n#define TC_BLOCK {int(block)}\n')
    if stats:
        w.append("#define TC_STATS 1\n")
    w.append("// solver configuration (reference: cmex2optimizeCS.m:303-331)\n")
    for k in ["nU", "nG", "nF", "scaleCost", "scaleInequalities", "scaleEqualities", "coupledAlphas",
              "setAddEye2Hessian", "adjustAddEye2Hessian", "useInertia", "delta", "skipAffine",
              "smallerNewtonMatrix", "useLDL", "useUmfpack", "allowSave"]:
        w.append(f"#define {k} {int(spec.defines[k])}\n")
    for k in ["gradTolerance", "addEye2HessianUtolerance", "equalTolerance", "desiredDualityGap",
              "alphaMin", "alphaMax", "muFactorAggressive", "muFactorConservative"]:
        w.append(f"#define {k} {_lit(spec.defines[k])}\n")
    nOut = sum(int(np.prod(sz)) if len(sz) else 1 for _, sz in spec.outputs)
    w.append(f"#define TC_NM {max(1, L.nPriv)}\n#define TC_NEXT {L.nExt}\n#define TC_NOUT {nOut}\n"
             f"#define TC_NGROUPS {T.nGroups}\n#define TC_NSTW {nW}\n\n")
    w.append("struct TcCtx {\n  double m[TC_NM];\n  const double *__restrict__ S;\n  long B, tid;\n"
             "  unsigned long long st[TC_NSTW];\n#ifdef TC_STATS\n  int nMinF;\n#endif\n};\n"
             "#define SX(k) (c.S[(long)(k) * c.B + c.tid])\n#define MM(k) (c.m[k])\n\n")
    # dependency groups ------------------------------------------------------------------
    for g, members in enumerate(T.groups, 1):
        w.append(f"// group {g}: {len(members)} instructions, {int(T.flops[g].sum())} flops\n")
        w.append(f"static __device__ __noinline__ void tc_grp{g}(TcCtx &c) {{\n")
        for i in members:
            w.append(L.statement(int(i)))
        w.append("}\n")
    w.append("\n#define TC_NEED(g) do { if (!((c.st[((g)-1)>>6] >> (((g)-1)&63)) & 1ull)) { "
             "c.st[((g)-1)>>6] |= 1ull << (((g)-1)&63); tc_grp##g(c); } } while (0)\n")

    def inval(children):
        s = ""
        for wd in range(nW):
            m = _mask([g for g in children if (g - 1) // 64 == wd]) >> (64 * wd) if False else \
                sum(1 << ((g - 1) % 64) for g in children if (g - 1) // 64 == wd)
            if m:
                s += f"  c.st[{wd}] &= ~0x{m:x}ull;\n"
        return s

    # entry points ------------------------------------------------------------------------
    w.append("\n// ---- tape entry points (semantics: writeCfunctionpergroup.m:513-852)\n")
    for e in T.sets:
        if e["external"]:
            continue
        w.append(f"static __device__ __forceinline__ void {e['name']}(TcCtx &c, const double *input) {{\n")
        for k, i in enumerate(e["ins"]):
            if L.pslot[i] >= 0:
                w.append(f"  MM({L.pslot[i]}) = input[{k}];\n")
        w.append(inval(e["children"]))
        w.append("}\n")
    for e in T.gets:
        args = ", ".join(f"double *y{k + 1}" for k in range(len(e["outputs"])))
        w.append(f"static __device__ __forceinline__ void {e['name']}(TcCtx &c, {args}) {{\n")
        for g in e["parents"]:
            w.append(f"  TC_NEED({g});\n")
        for k, o in enumerate(e["outputs"]):
            for j, i in enumerate(o["ins"]):
                w.append(f"  y{k + 1}[{j}] = {L.ref(i)};\n")
        w.append("}\n")
    for e in T.copies:
        w.append(f"static __device__ __forceinline__ void {e['name']}(TcCtx &c) {{\n")
        for g in e["parents"]:
            w.append(f"  TC_NEED({g});\n")
        for d, s in zip(e["dst"], e["src"]):
            if L.external[d]:
                raise NotImplementedError("copy into an external set")
            if L.pslot[d] >= 0:
                w.append(f"  MM({L.pslot[d]}) = {L.ref(s)};\n")
        w.append(inval(e["children"]))
        w.append("}\n")
    # entry points the loop may reference under #if but that this problem does not have
    w.append("\n#define TC_POW_HALF(x) sqrt(x)\n")
    w.append(f'#include "ipm_newton.hip.inc"\n\n')
    # kernels -------------------------------------------------------------------------------
    outs = T.entry(f"{cls}_getOutputs")[2]["outputs"]
    w.append("extern \"C\" __global__ void __launch_bounds__(TC_BLOCK)\n"
             "tc_solve_kernel(const double *__restrict__ S, double *__restrict__ O, int *__restrict__ status,\n"
             "                int *__restrict__ iter, double mu0, int maxIter, double ae0, double ae1, long B) {\n"
             "  const long tid = (long)blockIdx.x * TC_BLOCK + threadIdx.x;\n"
             "  if (tid >= B) return;\n"
             "  TcCtx c;\n  c.S = S;\n  c.B = B;\n  c.tid = tid;\n"
             "  for (int k = 0; k < TC_NSTW; k++) c.st[k] = 0;\n"
             "#ifdef TC_STATS\n  c.nMinF = 0;\n#endif\n"
             "  int st = 0, it = 0;\n"
             "  ipmPD_CSsolver(c, mu0, maxIter, ae0, ae1, &st, &it);\n")
    for k, o in enumerate(outs):
        w.append(f"  double y{k + 1}[{max(1, len(o['ins']))}];\n")
    w.append(f"  {cls}_getOutputs(c, " + ", ".join(f"y{k + 1}" for k in range(len(outs))) + ");\n")
    off = 0
    out_meta = []
    for k, o in enumerate(outs):
        n_ = len(o["ins"])
        w.append(f"  for (int j = 0; j < {n_}; j++) O[(long)({off} + j) * B + tid] = y{k + 1}[j];\n")
        out_meta.append((o["name"], o["size"], n_, off))
        off += n_
    w.append("  status[tid] = st;\n  iter[tid] = it;\n}\n\n")
    # probe kernel: init sequence then one get
    w.append("extern \"C\" __global__ void __launch_bounds__(TC_BLOCK)\n"
             "tc_probe_kernel(const double *__restrict__ S, double *__restrict__ out, int which, double mu0,\n"
             "                double ae0, double ae1, long B) {\n"
             "  const long tid = (long)blockIdx.x * TC_BLOCK + threadIdx.x;\n"
             "  if (tid >= B) return;\n"
             "  TcCtx c;\n  c.S = S;\n  c.B = B;\n  c.tid = tid;\n"
             "  for (int k = 0; k < TC_NSTW; k++) c.st[k] = 0;\n"
             "#if setAddEye2Hessian != 0\n  setAddEye2HessianU__(c, &ae0);\n  setAddEye2HessianEq__(c, &ae1);\n#endif\n"
             "  initPrimal__(c);\n"
             "#if scaleCost != 0\n  scaleCost__(c);\n#endif\n"
             "#if nF > 0\n#if scaleInequalities != 0\n  scaleIneq__(c);\n#endif\n  setMu__(c, &mu0);\n  initDualIneq__(c);\n#endif\n"
             "#if nG > 0\n  initDualEqX__(c);\n#endif\n"
             "  switch (which) {\n")
    probes = []
    for k, e in enumerate(T.gets):
        tot = sum(len(o["ins"]) for o in e["outputs"])
        probes.append((e["name"], tot))
        w.append(f"    case {k}: {{\n")
        for q, o in enumerate(e["outputs"]):
            w.append(f"      double y{q + 1}[{max(1, len(o['ins']))}];\n")
        w.append(f"      {e['name']}(c, " + ", ".join(f"y{q + 1}" for q in range(len(e["outputs"]))) + ");\n")
        po = 0
        for q, o in enumerate(e["outputs"]):
            n_ = len(o["ins"])
            w.append(f"      for (int j = 0; j < {n_}; j++) out[tid * {tot} + {po} + j] = y{q + 1}[j];\n")
            po += n_
        w.append("    } break;\n")
    w.append("    default: break;\n  }\n}\n\n")
    # metadata tables ---------------------------------------------------------------------------
    isvar = {nm for nm, _ in spec.variables}

    def carr(ctype, name, vals, fmt=str):
        return f"static const {ctype} {name}[] = {{{', '.join(fmt(v) for v in vals) or '0'}}};\n"

    def dims8(sz):
        d = list(sz) + [1] * (8 - len(sz))
        return "{" + ", ".join(str(int(x)) for x in d) + "}"

    ins = L.inputs
    w.append(f"static const int TC_NIN = {len(ins)};\n")
    w.append(carr("char *const", "TC_IN_NAME", [i["name"] for i in ins], lambda s: f'"{s}"'))
    w.append(carr("int", "TC_IN_ISVAR", [int(i["name"] in isvar) for i in ins]))
    w.append(carr("int64_t", "TC_IN_NUMEL", [i["numel"] for i in ins]))
    w.append(carr("int64_t", "TC_IN_OFF", [i["off"] for i in ins]))
    w.append(carr("int", "TC_IN_NDIM", [len(i["size"]) for i in ins]))
    w.append("static const int64_t TC_IN_DIMS[][8] = {" + ", ".join(dims8(i["size"]) for i in ins) + "};\n")
    w.append(f"static const int TC_NOUTS = {len(out_meta)};\n")
    w.append(carr("char *const", "TC_OUT_NAME", [o[0] for o in out_meta], lambda s: f'"{s}"'))
    w.append(carr("int64_t", "TC_OUT_NUMEL", [o[2] for o in out_meta]))
    w.append(carr("int64_t", "TC_OUT_OFF", [o[3] for o in out_meta]))
    w.append(carr("int", "TC_OUT_NDIM", [len(o[1]) for o in out_meta]))
    w.append("static const int64_t TC_OUT_DIMS[][8] = {" + ", ".join(dims8(o[1]) for o in out_meta) + "};\n")
    dn = sorted(spec.defines)
    w.append(f"static const int TC_NDEF = {len(dn)};\n")
    w.append(carr("char *const", "TC_DEF_NAME", dn, lambda s: f'"{s}"'))
    w.append(carr("double", "TC_DEF_VAL", [spec.defines[k] for k in dn], lambda v: repr(float(v))))
    w.append(f"static const int TC_NPROBE = {len(probes)};\n")
    w.append(carr("char *const", "TC_PROBE_NAME", [p[0] for p in probes], lambda s: f'"{s}"'))
    w.append(carr("int64_t", "TC_PROBE_NUMEL", [p[1] for p in probes]))
    stat = {"total_instructions": int(T.n), "used_instructions": int(T.used.sum()),
            "groups": int(T.nGroups), "private_doubles": int(L.nPriv), "input_doubles": int(L.nExt),
            "output_doubles": int(nOut), "flops_all_groups": int(T.flops.sum())}
    for g in range(1, T.nGroups + 1):
        stat[f"flops_group{g}"] = int(T.flops[g].sum())
    w.append(f"static const int TC_NSTAT = {len(stat)};\n")
    w.append(carr("char *const", "TC_STAT_NAME", list(stat), lambda s: f'"{s}"'))
    w.append(carr("int64_t", "TC_STAT_VAL", list(stat.values())))
    # reference-exact single-instance wrappers (B=1)
    cw = ""
    for i in ins:
        cw += (f'void {cls}_set_{i["name"]}(const double *input) '
               f'{{ tc_compat_set("{i["name"]}", input); }} \\\n')
    args = ", ".join(f"double *y{k + 1}" for k in range(len(out_meta)))
    cw += (f"void {cls}_getOutputs({args}) {{ double *o_[] = {{"
           + ", ".join(f"y{k + 1}" for k in range(len(out_meta))) + "}; tc_compat_get(o_); }\n")
    w.append("#define TC_COMPAT_WRAPPERS \\\n" + cw)
    w.append('\n#include "tc_runtime.hip.inc"\n')
    return "".join(w), L


def class_header(spec):
    """<cls>.h: the reference-exact C symbols (writeCfunctionpergroup.m:513-635)."""
    cls = spec.classname
    s = (f"/* GENERATED: single-instance C surface of class {cls}; same symbols the reference's\n"
         f" * generated {cls}.c exports (writeCfunctionpergroup.m:524,607-635; ipmPD_CSsolver.c:132-141).\n"
         f" * The batched interface is include/tenscalc_hip.h. */\n#include <stdint.h>\n"
         "#ifdef __cplusplus\nextern \"C\" {\n#endif\n")
    for nm, _ in spec.parameters + spec.variables:
        s += f"void {cls}_set_{nm}(const double *input);\n"
    s += ("void ipmPD_CSsolver(double *mu0, int32_t *maxIter, int32_t *saveIter, double *addEye2Hessian,\n"
          "                    int32_t *status, int32_t *iter, double *time);\n")
    s += f"void {cls}_getOutputs(" + ", ".join(f"double *y{k + 1}" for k in range(len(spec.outputs))) + ");\n"
    s += "#ifdef __cplusplus\n}\n#endif\n"
    return s


def build_class(spec, fp_contract="fast", arch="gfx950", opt="-O3", block=64, force=False,
                outdir=None, stats=False, tag=None, extra_flags=()):
    """Emit <cls>.hip (+ <cls>.h, <cls>.tape) and compile lib<cls>.so with hipcc for gfx950.
    Mirrors the createGateway/compile step of cmex2optimizeCS.m:403-417."""
    from .tapeio import save_tape
    cls = spec.classname
    outdir = outdir or os.path.join(spec.folder, f"@{cls}")
    os.makedirs(outdir, exist_ok=True)
    suffix = f"_{tag}" if tag else ""
    src, L = emit_class(spec, fp_contract=fp_contract, block=block, stats=stats)
    hip = os.path.join(outdir, f"{cls}{suffix}.hip")
    so = os.path.join(outdir, f"lib{cls}{suffix}.so")
    flags = [opt, f"--offload-arch={arch}", "-fPIC", "-shared", "-std=c++17",
             f"-ffp-contract={fp_contract}", "-I", CSRC, "-I", INCLUDE, "-Wno-unused-value"] + list(extra_flags)
    deps = [os.path.join(CSRC, "ipm_newton.hip.inc"), os.path.join(CSRC, "tc_runtime.hip.inc"),
            os.path.join(INCLUDE, "tenscalc_hip.h")]
    h = hashlib.sha256((src + " ".join(flags)).encode())
    for d in deps:
        h.update(open(d, "rb").read())
    stamp = os.path.join(outdir, f"{cls}{suffix}.stamp")
    digest = h.hexdigest()
    if (not force and os.path.exists(so) and os.path.exists(stamp)
            and open(stamp).read().strip() == digest):
        return so
    with open(hip, "w") as f:
        f.write(src)
    with open(os.path.join(outdir, f"{cls}.h"), "w") as f:
        f.write(class_header(spec))
    save_tape(spec.tape, os.path.join(outdir, f"{cls}.tape"))
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + flags + [hip, "-o", so]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError(f"hipcc failed for class {cls}:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
    with open(stamp, "w") as f:
        f.write(digest)
    return so
