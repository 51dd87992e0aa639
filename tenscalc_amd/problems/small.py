"""Small least-squares problems that exercise the branches of the solver template the BASELINE
problems do not: equality + inequality constraints with inertia-based regularisation and
uncoupled step lengths, equality constraints only (nF==0), and no constraints at all."""
from __future__ import annotations

from .. import tcalculus as tc
from ..cmex2optimizeCS import cmex2optimizeCS

KINDS = {
    "eq": dict(useInertia=True, coupledAlphas=False, scaleCost=10.0),
    "eqonly": dict(),
    "free": dict(adjustAddEye2Hessian=False),
}


def small_ls(kind, N=12, n=4, **options):
    A = tc.Tvariable("A", (N, n))
    b = tc.Tvariable("b", (N,))
    x = tc.Tvariable("x", (n,))
    J = tc.norm2(A * x - b) / N
    cons = {"eq": [tc.tsum(x) == .1, x >= 0], "eqonly": [tc.tsum(x) == .1], "free": []}[kind]
    opt = dict(KINDS[kind])
    opt.update(options)
    return cmex2optimizeCS(classname="tmp_small_" + kind, objective=J, optimizationVariables=[x],
                           constraints=cons, outputExpressions=[J, x], outputNames=["J", "x"],
                           parameters=[A, b], **opt)


# every unary opcode of the scalar ISA that the BASELINE problems do not reach, as OUTPUT expressions of a small
# box-constrained least-squares problem (evaluated once, by getOutputs): transcendental families, rounding, the
# derivative chains of sqrt / atan / srelu, I_componentwise (C text) -- tests/test_gpu_opcodes.py
OPCODE_FUNS = ["exp", "log", "sin", "minus_sin", "cos", "minus_cos", "round", "ceil", "floor", "abs", "sign", "sqrt",
               "Dsqrt", "DDsqrt", "sqr", "2times", "cube", "3sqr", "6times", "atan", "Datan", "DDatan", "srelu", "dsrelu",
               "ddsrelu", "relu", "heaviside", "inv", "minus_inv_sqr", "2_inv_cube"]


def opcode_zoo(N=12, n=4, classname="tmp_opcode_zoo", **options):
    A = tc.Tvariable("A", (N, n))
    b = tc.Tvariable("b", (N,))
    x = tc.Tvariable("x", (n,))
    J = tc.norm2(A * x - b) / N
    z = 40.0 * x - 1.0 + b[0:n]                 # spread arguments: roughly (-1, 2)
    pos = 40.0 * x + .25                         # positive arguments for log / sqrt / inverse families
    outs, names = [J, x], ["J", "x"]
    for f in OPCODE_FUNS:
        arg = pos if f in ("log", "sqrt", "Dsqrt", "DDsqrt", "inv", "minus_inv_sqr", "2_inv_cube") else z
        outs.append(tc.compose(arg, f))
        names.append("f_" + f.replace("2", "two").replace("3", "three").replace("6", "six"))
    outs.append(tc.componentwise(z, "fabs(%s)"))
    names.append("cw_fabs")
    outs.append(tc.componentwise(pos, "sqrt( %s )"))
    names.append("cw_sqrt")
    return cmex2optimizeCS(classname=classname, objective=J, optimizationVariables=[x],
                           constraints=[x >= 0, x <= .05], outputExpressions=outs, outputNames=names,
                           parameters=[A, b], **options)


def sensitivity_ls(N=12, n=3, ns=2, classname="tmp_sens", **options):
    """constrained least squares in z = (p, x) with p declared as SENSITIVITY variables (cmex2optimizeCS's
    ``sensitivityVariables``, lib/ipmPD_CS.m:478-503): outputs Du1 = d(x, nu, lambda)/dp, DfDu1, D2fDu1 next to the
    solver's Hess and Lf_u"""
    A = tc.Tvariable("A", (N, ns + n))
    b = tc.Tvariable("b", (N,))
    x = tc.Tvariable("x", (n,))
    p = tc.Tvariable("p", (ns,))
    J = tc.norm2(A * tc.cat(1, p, x) - b) / N
    nZ = ns + n + 1 + n
    outs = [J, x, p, tc.Tvariable("Hess_", (nZ, nZ)), tc.Tvariable("Lf_u_", (ns + n,)), tc.Tvariable("Du1_", (nZ - ns, ns)),
            tc.Tvariable("DfDu1_", (ns,)), tc.Tvariable("D2fDu1_", (ns, ns))]
    return cmex2optimizeCS(classname=classname, objective=J, optimizationVariables=[p, x], sensitivityVariables=[p],
                           constraints=[tc.tsum(x) == .1, x >= 0], outputExpressions=outs,
                           outputNames=["J", "x", "p", "Hess", "Lf_u", "Du1", "DfDu1", "D2fDu1"], parameters=[A, b],
                           lambdaUpdate="additive", **options)


def long_reductions(M=600, N=12, n=4, classname="tmp_long_reductions", **options):
    """long minimum / maximum / infinity norm / clp / inner product over parameters of M entries, as OUTPUT expressions of a
    small box-constrained least-squares problem (evaluated once, by getOutputs; the solver never sees v, w): what the tile
    machine evaluates as parallel parts + a combining record (interp_schedule.split_long) -- tests/test_gpu_opcodes.py feeds
    them NaNs, signed zeros and infinities"""
    A = tc.Tvariable("A", (N, n))
    b = tc.Tvariable("b", (N,))
    x = tc.Tvariable("x", (n,))
    v = tc.Tvariable("v", (M,))
    w = tc.Tvariable("w", (M,))
    J = tc.norm2(A * x - b) / N
    outs = [J, x, tc.tmin(v), tc.tmax(v), tc.norminf(v), tc.clp(v, w), tc.tprod(v, [-1], w, [-1])]
    return cmex2optimizeCS(classname=classname, objective=J, optimizationVariables=[x],
                           constraints=[x >= 0, x <= .05], outputExpressions=outs,
                           outputNames=["J", "x", "vmin", "vmax", "vinf", "vclp", "vdotw"], parameters=[A, b, v, w], **options)
