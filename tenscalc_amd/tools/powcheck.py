"""Does THIS machine's libm return what the device returns?  The one libm call on the hot path is `pow`:
`.5*pow(x,-.5)` / `-.25*pow(x,-1.5)` of I_Dsqrt / I_DDsqrt (instructionsTableUTHash.c:1418,1424) and the Newton loops'
`pow(mu,.5)` (ipmPD_CSsolver.c:801, ipmPDeq_CSsolver.c:692).  The reference's result is whatever libm its generated C
links; the device restates glibc 2.35's `__pow_fma` operation by operation (csrc/tc_glibcpow.hip.inc, tables read from
that libm).  On a host with another glibc, or without FMA3, the two differ in the last bit on a fraction of arguments
and "bit-exact" would be a claim nobody checked -- so the claim is checked where it is made: `bench.py` and
`__graft_entry__.smoke()` call `pow_parity()` before they print a parity class, and downgrade it on any mismatch.

build_host / build_dev compile the comparison harness (tools/crmath/*) for the host (gcc) and for gfx950 (hipcc)."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "crmath")
CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "csrc")
DOWNGRADED = "solver-tolerance (host libm differs)"


def build_host():
    """gcc build of the device's pow files for the host, next to the host libm's pow -> ctypes library"""
    out = os.path.join(HERE, "libcrmath_host.so")
    srcs = [os.path.join(HERE, "crmath_host.c"), os.path.join(CSRC, "tc_crmath.hip.inc"),
            os.path.join(CSRC, "tc_glibcpow.hip.inc"), os.path.join(CSRC, "tc_glibcpow_tables.inc")]
    if not os.path.exists(out) or os.path.getmtime(out) < max(map(os.path.getmtime, srcs)):
        tmp = out + ".%d.tmp" % os.getpid()
        subprocess.run(["gcc", "-O1", "-ffp-contract=off", "-std=gnu99", "-fPIC", "-shared", "-I", CSRC, srcs[0], "-o", tmp,
                        "-lm"], check=True)
        os.replace(tmp, out)
    return ctypes.CDLL(out)


def build_dev():
    """hipcc build of the same files for gfx950 (cross-compiles without a GPU; __graft_entry__.build() calls this)"""
    out = os.path.join(HERE, "libcrmath_dev.so")
    srcs = [os.path.join(HERE, "crmath_dev.hip")] + [os.path.join(CSRC, f) for f in
            ("tc_pow.hip.inc", "tc_crmath.hip.inc", "tc_glibcpow.hip.inc", "tc_glibcpow_tables.inc")]
    if not os.path.exists(out) or os.path.getmtime(out) < max(map(os.path.getmtime, srcs)):
        tmp = out + ".%d.tmp" % os.getpid()
        subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-O3", "--offload-arch=gfx950", "-std=c++17",
                        "-ffp-contract=off", "-fPIC", "-shared", "-I", CSRC, srcs[0], "-o", tmp], check=True)
        os.replace(tmp, out)
    return out


def eval2(fn, x):
    """call one of the harness's (x, out1, out2, n) functions"""
    x = np.ascontiguousarray(x, dtype=np.float64)
    a, b = np.empty_like(x), np.empty_like(x)
    dp = ctypes.POINTER(ctypes.c_double)
    fn(x.ctypes.data_as(dp), a.ctypes.data_as(dp), b.ctypes.data_as(dp), ctypes.c_long(x.size))
    return a, b


def libc_version():
    try:
        f = ctypes.CDLL(None).gnu_get_libc_version
        f.restype = ctypes.c_char_p
        return "glibc " + f().decode()
    except Exception:            # not glibc
        return "unknown libc"


def sample(n, seed):
    """arguments a solve produces: barrier parameters and squared norms over many binades"""
    r = np.random.RandomState(seed)
    return np.concatenate([r.uniform(1e-3, 500, n // 2), np.exp(r.uniform(-60, 60, n // 4)), r.uniform(0.5, 2.0, n - n // 2 - n // 4)])


def _same(u, v):
    return int((~((u == v) | (np.isnan(u) & np.isnan(v)))).sum())


def host_pow(x):
    """this machine's pow(x,-.5), pow(x,-1.5), pow(x,.5) -- live calls of the libm the reference's generated C would link"""
    L = build_host()
    a, b = eval2(L.libm_eval, x)
    _, h = eval2(L.half_eval, x)
    return a, b, h


def pow_parity(n=200_000, device=None, seed=2025):
    """Compare what the solver kernels compute -- TC_POW_NEG(x,-.5), TC_POW_NEG(x,-1.5), TC_POW_HALF(x) -- with live calls of
    this host's pow on n arguments.  device=True: the gfx950 build on the GPU (what runs in the kernels); False: the same
    files compiled for the host (no GPU needed; the -m gpu test ties the two together); None: the GPU when torch sees one.
    -> dict(ok, neg_half / neg_three_halves / half = mismatch counts, n, on, libc)"""
    x = sample(int(n), seed)
    pa, pb, ph = host_pow(x)
    if device is None:
        try:
            import torch
            device = bool(torch.cuda.is_available())
        except Exception:
            device = False
    if device:
        D = ctypes.CDLL(build_dev())
        dp = ctypes.POINTER(ctypes.c_double)
        o = [np.empty_like(x) for _ in range(4)]
        rc = D.crm_dev_eval(x.ctypes.data_as(dp), *[v.ctypes.data_as(dp) for v in o], ctypes.c_long(x.size))
        if rc:
            raise RuntimeError("pow_parity: HIP error %d in the device harness" % rc)
        da, db, dh = o[0], o[1], o[3]
    else:
        L = build_host()
        da, db = eval2(L.glibc_emul_eval, x)
        dh, _ = eval2(L.half_eval, x)
    res = dict(neg_half=_same(da, pa), neg_three_halves=_same(db, pb), half=_same(dh, ph), n=int(x.size),
               on="gfx950" if device else "host build of the device's files", libc=libc_version())
    res["ok"] = res["neg_half"] == 0 and res["neg_three_halves"] == 0 and res["half"] == 0
    return res


def tape_calls_pow(tape, defines=None):
    """which of the device's pow replacements a class depends on: (TC_POW_NEG via I_Dsqrt / I_DDsqrt, TC_POW_HALF via the
    mu update of a loop with inequality constraints, ipmPD_CSsolver.c:801 under nF > 0)"""
    from .. import instructions as I
    used = tape.used[1:] if hasattr(tape, "used") else np.ones(tape.n, dtype=bool)
    neg = bool(np.any(((tape.types == I.I_Dsqrt) | (tape.types == I.I_DDsqrt)) & used))
    half = bool((defines or tape.defines).get("nF", 0) > 0)
    return neg, half


LIBM_FUNCTIONS = ("exp", "log", "sin", "cos", "atan")        # order = the harness's `which`


def tape_calls_libm(tape):
    """which of exp / log / sin / cos / atan a class evaluates (I_exp, I_log, I_sin, I_minus_sin, I_cos, I_minus_cos, I_atan,
    the srelu family, I_componentwise instructions that are a plain call of one of them)"""
    from .. import instructions as I
    used = tape.used[1:] if hasattr(tape, "used") else np.ones(tape.n, dtype=bool)
    by_op = {I.I_exp: {"exp"}, I.I_log: {"log"}, I.I_sin: {"sin"}, I.I_minus_sin: {"sin"}, I.I_cos: {"cos"}, I.I_minus_cos: {"cos"},
             I.I_atan: {"atan"}, I.I_srelu: {"exp", "log"}, I.I_dsrelu: {"exp"}, I.I_ddsrelu: {"exp"}}
    out = set()
    for op, names in by_op.items():
        if np.any((tape.types == op) & used):
            out |= names
    for i in np.flatnonzero((tape.types == I.I_componentwise) & used):
        eq = I.componentwise_opcode(tape.pars[tape.parStart[i]:tape.parStart[i + 1]])
        out |= by_op.get(eq, set())
    return sorted(out)


def libm_parity(names, n=200_000, device=None, seed=2026):
    """the device's TC_EXP .. TC_ATAN (csrc/tc_pow.hip.inc) against live calls of this host's libm on n arguments each
    -> {name: mismatches}"""
    r = np.random.RandomState(seed)
    args = {"exp": r.uniform(-700, 700, n), "log": np.exp(r.uniform(-700, 700, n)), "sin": r.uniform(-1e3, 1e3, n) * 10.0 ** r.randint(-6, 6, n),
            "cos": r.uniform(-1e3, 1e3, n) * 10.0 ** r.randint(-6, 6, n), "atan": r.uniform(-50, 50, n) * 10.0 ** r.randint(-6, 3, n)}
    if device is None:
        try:
            import torch
            device = bool(torch.cuda.is_available())
        except Exception:
            device = False
    L = build_host()
    D = ctypes.CDLL(build_dev()) if device else None
    dp = ctypes.POINTER(ctypes.c_double)
    out = {}
    for nm in names:
        w = LIBM_FUNCTIONS.index(nm)
        x = np.ascontiguousarray(args[nm])
        want, got = np.empty_like(x), np.empty_like(x)
        L.unary_libm_eval(ctypes.c_int(w), x.ctypes.data_as(dp), want.ctypes.data_as(dp), ctypes.c_long(x.size))
        if D is not None:
            if D.crm_dev_unary(ctypes.c_int(w), x.ctypes.data_as(dp), got.ctypes.data_as(dp), ctypes.c_long(x.size)):
                raise RuntimeError("libm_parity: HIP error in the device harness")
        else:
            L.unary_emul_eval(ctypes.c_int(w), x.ctypes.data_as(dp), got.ctypes.data_as(dp), ctypes.c_long(x.size))
        out[nm] = _same(got, want)
    return out


def checked_parity_class(label, tape, defines=None, check=None):
    """the parity class to PRINT: `label` (__graft_entry__.PARITY_CLASS) unless it says bit-exact, the class depends on
    libm (pow; exp / log / sin / cos / atan), and this host's libm is not what the device restates -> (label, check dict)"""
    if label != "bit-exact":
        return label, None
    neg, half = tape_calls_pow(tape, defines)
    fns = tape_calls_libm(tape)
    if not (neg or half or fns):
        return label, None
    bad = False
    if neg or half:
        check = check if check is not None else pow_parity()
        bad = bool((neg and (check["neg_half"] or check["neg_three_halves"])) or (half and check["half"]))
    if fns:
        lm = libm_parity(fns)
        check = dict(check or {}, libm=lm)
        bad = bad or any(lm.values())
    return (DOWNGRADED if bad else label), check
