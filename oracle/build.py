"""Recipes that build the checker: the C oracle (gcc) and, when /root/reference is present, the
reference's own Newton loop linked against the oracle's tape entry points (oracle/_ref/).

Outputs: oracle/libtc_oracle.so and oracle/_ref/libipmref_<tag>.so (both git-ignored, both
travel to the GPU box).  No reference source is copied: ipmPD_CSsolver.c is compiled from
/root/reference/lib where it lies, with the -D block of cmex2optimizeCS.m:303-331.
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SOLVER = "/root/reference/lib/ipmPD_CSsolver.c"
REF_SOLVER_EQ = "/root/reference/lib/ipmPDeq_CSsolver.c"
# the -D block of the equilibrium solvers (cmex2equilibriumLatentCS.m:417-448)
EQ_DEFINE_NAMES = ["nZ", "nU", "nD", "nX", "nG", "nF", "nNu", "gradTolerance", "addEye2Hessian1tolerance", "equalTolerance",
                   "desiredDualityGap", "scaleCost", "scaleInequalities", "scaleEqualities", "alphaMin", "alphaMax",
                   "coupledAlphas", "setAddEye2Hessian", "adjustAddEye2Hessian", "muFactorAggressive",
                   "muFactorConservative", "delta", "skipAffine", "smallerNewtonMatrix", "useLDL", "useUmfpack",
                   "allowSave", "debugConvergence", "debugConvergenceThreshold", "profiling"]


def _run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError("command failed: %s\n%s\n%s" % (" ".join(cmd), r.stdout, r.stderr))


def build_oracle(opt="-O1"):
    """-O1 is the reference's "fastest run-time" setting (examples/l1l2estimationCS.m:22-25).
    -ffp-contract=off: the reference build targets plain x86-64 (no FMA)."""
    san = os.environ.get("TC_SANITIZE")      # scripts/sanitize_cpu.sh: AddressSanitizer + UBSan build of the checker
    out = os.path.join(HERE, "libtc_oracle_san.so" if san else "libtc_oracle.so")
    srcs = [os.path.join(HERE, "tc_oracle.c"), os.path.join(HERE, "tc_oracle_eq.c")]
    if san:
        _run(["gcc", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-ffp-contract=off", "-std=gnu99",
              "-fPIC", "-shared", "-I", HERE] + srcs + ["-o", out, "-lm"])
        return out
    if os.path.exists(out) and os.path.getmtime(out) >= max(
            [os.path.getmtime(s_) for s_ in srcs] + [os.path.getmtime(os.path.join(HERE, "tc_oracle.h"))]):
        return out
    _run(["gcc", opt, "-ffp-contract=off", "-std=gnu99", "-fPIC", "-shared", "-I", HERE] + srcs + ["-o", out, "-lm"])
    return out


def _fmt(v):
    if isinstance(v, float):
        return "%e" % v          # cmex2optimizeCS.m:307-320 prints doubles with sprintf('%e')
    return str(int(v))


def build_ref(tag, defines, opt="-O1", solver="ipmPD_CSsolver"):
    """Compile the reference Newton loop (lib/ipmPD_CSsolver.c, or lib/ipmPDeq_CSsolver.c for solver="ipmPDeq_CSsolver")
    for one problem's #define block; returns the .so path or None when /root/reference is absent (GPU box)."""
    out = os.path.join(HERE, "_ref", "libipmref_%s.so" % tag)
    ref = REF_SOLVER_EQ if solver == "ipmPDeq_CSsolver" else REF_SOLVER
    if not os.path.exists(ref):
        return out if os.path.exists(out) else None
    os.makedirs(os.path.join(HERE, "_ref"), exist_ok=True)
    names = EQ_DEFINE_NAMES if solver == "ipmPDeq_CSsolver" else ["nU", "nG", "nF", "gradTolerance", "addEye2HessianUtolerance", "equalTolerance",
             "desiredDualityGap", "scaleCost", "scaleInequalities", "scaleEqualities", "alphaMin",
             "alphaMax", "coupledAlphas", "setAddEye2Hessian", "adjustAddEye2Hessian", "useInertia",
             "muFactorAggressive", "muFactorConservative", "delta", "skipAffine",
             "smallerNewtonMatrix", "useLDL", "useUmfpack", "allowSave", "debugConvergence",
             "profiling"]
    dflags = ["-D%s=%s" % (k, _fmt(defines[k])) for k in names]
    dflags += ["-DverboseLevel=0", "-DEXPORT=", '-DsaveNamePrefix="x"']
    # intermediates and the library under process-private names, the library renamed into place: two test workers
    # (pytest -n) asking for the same tag at the same time must not see each other's half-written files
    obj = out[:-3] + ".%d.o" % os.getpid()
    tmp = out + ".%d.tmp" % os.getpid()
    # the -D block only applies to the reference file; the glue and interpreter get none
    _run(["gcc", "-w", opt, "-ffp-contract=off", "-std=gnu99", "-fPIC", "-c"] + dflags + [ref, "-o", obj])
    _run(["gcc", opt, "-ffp-contract=off", "-std=gnu99", "-fPIC", "-shared", "-I", HERE, obj,
          os.path.join(HERE, "ref_glue.c"), os.path.join(HERE, "tc_oracle.c"), os.path.join(HERE, "tc_oracle_eq.c"),
          "-o", tmp, "-lm"])
    os.remove(obj)
    os.replace(tmp, out)
    return out


DEFINE_NAMES = ["nU", "nG", "nF", "gradTolerance", "addEye2HessianUtolerance", "equalTolerance",
                "desiredDualityGap", "scaleCost", "scaleInequalities", "scaleEqualities", "alphaMin",
                "alphaMax", "coupledAlphas", "setAddEye2Hessian", "adjustAddEye2Hessian", "useInertia",
                "muFactorAggressive", "muFactorConservative", "delta", "skipAffine",
                "smallerNewtonMatrix", "useLDL", "useUmfpack", "allowSave"]


def build_devloop(tag, defines, opt="-O1", solver="ipmPD_CSsolver"):
    """Host build (g++) of the device Newton loop tenscalc_amd/csrc/ipm_newton.hip.inc (ipm_newton_eq.hip.inc for
    solver="ipmPDeq_CSsolver") over the oracle's tape interpreter (oracle/devloop_glue.cpp), one library per #define block."""
    out = os.path.join(HERE, "_ref", "libdevloop_%s.so" % tag)
    os.makedirs(os.path.join(HERE, "_ref"), exist_ok=True)
    csrc = os.path.join(os.path.dirname(HERE), "tenscalc_amd", "csrc")
    obj = out[:-3] + ".%d.o" % os.getpid()          # process-private names, see build_ref
    defs = out[:-3] + ".%d_defs.h" % os.getpid()
    tmp = out + ".%d.tmp" % os.getpid()
    with open(defs, "w") as f:
        f.write("".join("#define %s %s\n" % (k, _fmt(defines[k]))
                        for k in (EQ_DEFINE_NAMES if solver == "ipmPDeq_CSsolver" else DEFINE_NAMES)))
        if int(defines.get("verboseLevel", 0)) >= 3 and solver != "ipmPDeq_CSsolver":     # the recorded iteration table
            f.write("#define verboseLevel %d\n" % int(defines["verboseLevel"]))
    _run(["g++", "-w", opt, "-ffp-contract=off", "-fPIC", "-c", "-I", HERE, "-I", csrc] +
         (["-DTC_LOOP_EQ=1"] if solver == "ipmPDeq_CSsolver" else []) +
         ['-DTC_DEFINES_H="%s"' % defs, os.path.join(HERE, "devloop_glue.cpp"), "-o", obj])
    _run(["gcc", opt, "-ffp-contract=off", "-std=gnu99", "-fPIC", "-c", "-I", HERE,
          os.path.join(HERE, "tc_oracle.c"), "-o", obj[:-2] + "_o.o"])
    _run(["gcc", opt, "-ffp-contract=off", "-std=gnu99", "-fPIC", "-c", "-I", HERE,
          os.path.join(HERE, "tc_oracle_eq.c"), "-o", obj[:-2] + "_e.o"])
    _run(["g++", "-shared", obj, obj[:-2] + "_o.o", obj[:-2] + "_e.o", "-o", tmp, "-lm"])
    for f_ in (obj, obj[:-2] + "_o.o", obj[:-2] + "_e.o", defs):
        os.remove(f_)
    os.replace(tmp, out)
    return out


if __name__ == "__main__":
    print(build_oracle())
