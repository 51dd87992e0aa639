"""Solver generator front door: ``cmex2optimizeCS`` with a HIP back-end.

Mirrors /root/reference/lib/cmex2optimizeCS.m (parameter names and meaning from
lib/private/parameters4optimize.m:12-168 and parameters4all.m:15-630): takes an objective,
optimization variables, constraints, parameters and output expressions written with
:mod:`tenscalc_amd.tcalculus`, builds the primal-dual interior-point tape and returns a
:class:`SolverSpec`.  ``SolverSpec.build()`` lowers the tape to a HIP class library
(:mod:`tenscalc_amd.hip_emitter`) and ``SolverSpec.load()`` returns an object with the
reference class surface ``setP_*/setV_*/solve/getOutputs`` extended by a trailing batch
dimension.
"""
from __future__ import annotations

import re

from . import tcalculus as tc
from .csparse import CSparse
from .ipm_assembly import ipmPD_CS, ipmPD_CStimesLambda
from .tcalculus import Tvariable, Tzeros, cat, reshape, substitute

__all__ = ["cmex2optimizeCS", "SolverSpec"]


def packExpressions(exprs):
    """lib/packExpressions.m"""
    if not exprs:
        return Tzeros((0,))
    return cat(1, *[reshape(e, (tc._prod(e.size),)) for e in exprs])


def packVariables(variables, name, *exprs):
    """lib/packVariables.m: one column vector replaces the listed variables."""
    n = 0
    where = []
    for v in variables:
        if v.op != "variable":
            raise ValueError(f"Can only pack variables (not {v.op})")
        ln = tc._prod(v.size)
        where.append((n, n + ln))
        n += ln
    out = Tvariable(name, (n,))
    repl = [reshape(out[a:b], v.size) for v, (a, b) in zip(variables, where)]
    res = [substitute(e, list(variables), repl) for e in exprs]
    return (out, where) + tuple(res)


def variableIndices(u, optimizationVariables, where, sensitivityVariables):
    """lib/private/variableIndices.m: flags, over the packed vector u, of the optimization variables (or of the
    selected entries v[...] of one) with respect to which the cost sensitivity is wanted"""
    import numpy as np
    isSensitivity = np.zeros(u.size[0], dtype=bool)
    for k, sv in enumerate(sensitivityVariables):
        sv = tc.toT(sv)
        ind = None
        if sv.op == "reshape" and sv.args[0].op == "subsref":       # v[i, :] : the dropped dimension is a reshape here
            sv = sv.args[0]
        if sv.op == "subsref":
            base = sv.args[0]
            lin = np.arange(tc._prod(base.size)).reshape(base.size, order="F")
            ind = lin[np.ix_(*[list(x) for x in sv.par])].ravel(order="F")
            sv = base
        if sv.op != "variable":
            raise ValueError("all sensitivityVariables must be of the type 'variable' (%dth is of type '%s')" % (k + 1, sv.op))
        for v, (a, b) in zip(optimizationVariables, where):
            if v.op == "variable" and v.par == sv.par:
                pos = np.arange(a, b)
                isSensitivity[pos if ind is None else pos[ind]] = True
                break
        else:
            raise ValueError(f"sensitivityVariable {sv.par} is not an optimizationVariable")
    return isSensitivity


def parseConstraints(code, classname, constraints):
    """lib/private/parseConstraints.m:74-135"""
    Gc, Fc, nus, lambdas = [], [], [], []
    for k, c in enumerate(constraints):
        if not isinstance(c, tc.Tcalculus):
            raise ValueError(f"constraint {k + 1} is not a Tcalculus expression")
        if c.op == "iszero":
            op1 = c.args[0]
            Gc.append(op1)
            v = Tvariable(f"nu{len(nus) + 1}__", op1.size)
            nus.append(v)
            code.declareSet(v, f"{classname}_set_{v.par}")
        elif c.op == "ispositive":
            op1 = c.args[0]
            Fc.append(op1)
            v = Tvariable(f"lambda{len(lambdas) + 1}__", op1.size)
            lambdas.append(v)
            code.declareSet(v, f"{classname}_set_{v.par}")
        else:
            raise ValueError(f"constraint of type '{c.op}' not implemented")
    return packExpressions(Gc), packExpressions(Fc), nus, lambdas


class SolverSpec:
    """Everything a back-end needs: the tape, the solver ``#define`` block
    (cmex2optimizeCS.m:303-331) and the class template (names, sizes)."""

    def __init__(self, classname, tape, defines, parameters, variables, outputs, folder):
        self.classname = classname
        self.tape = tape
        self.defines = defines
        self.parameters = parameters      # [(name, size)]
        self.variables = variables        # [(name, size)]
        self.outputs = outputs            # [(name, size)]
        self.folder = folder

    def build(self, **kw):
        from .hip_emitter import build_class
        return build_class(self, **kw)

    def load(self, **kw):
        from .runtime import load_class
        return load_class(self, **kw)


def cmex2optimizeCS(*, classname, objective, optimizationVariables, constraints=(),
                    parameters=(), outputExpressions=(), outputNames=None, folder=".",
                    gradTolerance=1e-4, equalTolerance=1e-4, desiredDualityGap=1e-5,
                    scaleInequalities=True, scaleCost=0.0, scaleEqualities=False,
                    muFactorAggressive=0.2, muFactorConservative=0.95, skipAffine=True,
                    delta=3, alphaMin=1e-7, alphaMax=1.0, coupledAlphas=True,
                    addEye2Hessian=True, adjustAddEye2Hessian=True, useInertia=False,
                    addEye2HessianUtolerance=1e-6, smallerNewtonMatrix=False, useLDL=True,
                    useUmfpack=False, allowSave=False, solverVerboseLevel=0, hoist=False, lambdaUpdate="multiplicative",
                    sensitivityVariables=(), ldlOrder=None, **ignored):
    """Generate an interior-point solver for ``min objective s.t. constraints``.

    ``lambdaUpdate='additive'`` generates the reference's original assembly lib/ipmPD_CS.m (``switch 1`` at
    cmex2optimizeCS.m:84-93) instead of ipmPD_CStimesLambda.m; only with it are ``skipAffine=False`` (Mehrotra
    predictor step, ipmPD_CSsolver.c:576-665) and ``smallerNewtonMatrix=True`` available.

    ``sensitivityVariables`` (optimization variables, or ``v[...]`` entries of one: lib/private/variableIndices.m)
    adds the cost-sensitivity outputs of ipmPD_CS.m:478-503 -- ``Du1_``, ``DfDu1_``, ``D2fDu1_`` placeholders in
    ``outputExpressions``; additive assembly only, as in the reference.

    ``ldlOrder`` (not a reference option; the reference calls symamd, sparsity_ldl.m:91) picks the elimination order of
    the KKT factorisation: ``"mindeg"`` (greedy minimum degree, the default) or ``("depth", width)`` -- the order
    with the shortest dependency chain that ``width`` parallel lanes can pay for in fill
    (csparse.depth_aware_order): same solution up to rounding, a several times shorter critical path on the
    time-banded KKT systems of the MPC / estimation examples.

    ``hoist=True`` (not a reference option) applies tenscalc_amd/hoist.py: products of
    parameter-only factors in the derivatives are contracted first, so they are evaluated once
    per solve instead of once per Newton iteration.  Same mathematics, different rounding.

    Unsupported reference options raise, exactly as ipmPD_CStimesLambda.m:31-48 does; unknown (misspelled)
    options raise too, as FunParTools' setParameters does for the reference."""
    if ignored:
        accepted_noop = {"compilerOptimization", "compileGateways", "compileLibrary", "compileStandalones",
                         "callType", "simulinkLibrary", "dummyLibrary", "pedigreeClass",
                         "executeScript", "useSparsityPatterns", "minInstructions4loop", "maxInstructionsPerFunction",
                         "absolutePath", "serverProgramName", "serverAddress", "port", "profiling", "umfpack",
                         "debugConvergence", "debugConvergenceThreshold", "scratchbookType", "fastRedundancyCheck",
                         "codeType", "targetComputer", "serverComputer", "parametersStructure", "parametersMfile"}
        bad = sorted(set(ignored) - accepted_noop)
        if bad:
            raise TypeError("cmex2optimizeCS: unknown option(s) %s" % ", ".join(bad))
    if lambdaUpdate not in ("multiplicative", "additive"):
        raise ValueError("lambdaUpdate: 'multiplicative' (ipmPD_CStimesLambda.m, what cmex2optimizeCS.m:92 selects) "
                         "or 'additive' (ipmPD_CS.m, cmex2optimizeCS.m:86)")
    if lambdaUpdate == "multiplicative":
        if not skipAffine:
            raise NotImplementedError("skipAffine=false needs lambdaUpdate='additive' (ipmPD_CStimesLambda.m:37-39)")
        if smallerNewtonMatrix:
            raise NotImplementedError("smallerNewtonMatrix=true needs lambdaUpdate='additive' (ipmPD_CStimesLambda.m:34-36)")
    if not useLDL or useUmfpack:
        raise NotImplementedError("only the LDL path is implemented (ipmPD_CStimesLambda.m:40-45)")
    classname = re.sub(r"[-+]", "_", classname)
    objective = tc.toT(objective)
    if objective.size != ():
        raise ValueError(f"Minimization criterion must be scalar (not {list(objective.size)})")
    optimizationVariables = list(optimizationVariables)
    parameters = list(parameters)
    for v in optimizationVariables:
        if not isinstance(v, tc.Tcalculus) or v.op != "variable":
            raise ValueError("all optimizationVariables must be of the type 'variable'")
        for p in parameters:
            if p.par == v.par:
                raise ValueError(f"optimization variable '{v.par}' cannot also be a parameter")
    outputExpressions = [tc.toT(o) for o in outputExpressions]
    if outputNames is None:
        outputNames = [o.par if o.op == "variable" else f"y{k + 1}"
                       for k, o in enumerate(outputExpressions)]

    code = CSparse()
    code.ldl_order = ldlOrder
    for p in parameters:
        code.declareSet(p, f"{classname}_set_{p.par}", external=True)
    for v in optimizationVariables:
        code.declareSet(v, f"{classname}_set_{v.par}", external=True)
    G, F, nus, lambdas = parseConstraints(code, classname, list(constraints))
    u, where, objective_u, outs_u, F, G = packVariables(
        optimizationVariables, "x_", objective, outputExpressions, F, G)
    outputExpressions = list(outs_u)
    u0 = packExpressions(optimizationVariables)
    code.declareCopy(u, u0, "initPrimal__")
    nG = G.size[0]
    nF = F.size[0]
    if nG > 0:
        nu, _, outputExpressions = packVariables(nus, "nu_", outputExpressions)
        outputExpressions = list(outputExpressions)
    else:
        nu = Tzeros((0,))
    if nF > 0:
        lam, _, outputExpressions = packVariables(lambdas, "lambda_", outputExpressions)
        outputExpressions = list(outputExpressions)
    else:
        lam = Tzeros((0,))
    rewrite = None
    if hoist:
        from .hoist import Hoister
        rewrite = Hoister(parameters).rewrite
    isSensitivity = variableIndices(u, optimizationVariables, where, list(sensitivityVariables))
    if isSensitivity.any() and lambdaUpdate != "additive":
        raise NotImplementedError("sensitivityVariables need lambdaUpdate='additive': ipmPD_CStimesLambda: isSensitivity=true "
                                  "not implemented, use ipmPD_CS.m (ipmPD_CStimesLambda.m:31-33)")
    if lambdaUpdate == "additive":
        Tout = ipmPD_CS(code, u, objective_u, F, G, lam, nu, addEye2Hessian=addEye2Hessian, scaleCost=scaleCost,
                        scaleInequalities=scaleInequalities, classname=classname, rewrite=rewrite,
                        skipAffine=bool(skipAffine) or nF == 0, smallerNewtonMatrix=bool(smallerNewtonMatrix),
                        isSensitivity=isSensitivity)
    else:
        Tout = ipmPD_CStimesLambda(code, u, objective_u, F, G, lam, nu,
                                   addEye2Hessian=addEye2Hessian, scaleCost=scaleCost,
                                   scaleInequalities=scaleInequalities, classname=classname, rewrite=rewrite)
    # outputs may refer to internal quantities through '<field>_' variables (cmex2optimizeCS.m:265-271)
    for fn, val in Tout.items():
        ph = Tvariable(fn + "_", val.size)
        outputExpressions = [substitute(o, ph, val) for o in outputExpressions]
    # the reference returns structurally sparse <=2-D outputs as sparse mxArrays
    # (writeCfunctionpergroup.m:590-605,716-727); the batched interface always returns dense arrays
    if rewrite is not None:
        outputExpressions = [rewrite(o) for o in outputExpressions]
    outputExpressions = [tc.full(o) for o in outputExpressions]
    code.declareGet(outputExpressions, f"{classname}_getOutputs", list(outputNames))
    if allowSave:        # ipmPD_CStimesLambda.m:283-291: declareSave(code,WW,'saveWW__',<cls>_WW.subscripts).  Here a get
        # of the Newton matrix that HipSolver.saveWW evaluates after the initialisation sequence (tenscalc_amd/typical.py)
        code.declareGet([tc.full(Tout["Hess"])], "saveWW__", ["WW"])

    # the reference passes the real-valued options to the C template through sprintf('%e') (cmex2optimizeCS.m:303-331):
    # six significant digits.  The same rounding here, so that a non-default value such as muFactorAggressive=1/3
    # gives the solver the reference would have generated (1e-4, .2, .95, ... are unchanged by it)
    pe = lambda v: float("%e" % float(v))          # noqa: E731
    gradTolerance, addEye2HessianUtolerance, equalTolerance, desiredDualityGap = (
        pe(gradTolerance), pe(addEye2HessianUtolerance), pe(equalTolerance), pe(desiredDualityGap))
    alphaMin, alphaMax, muFactorAggressive, muFactorConservative = (
        pe(alphaMin), pe(alphaMax), pe(muFactorAggressive), pe(muFactorConservative))
    defines = dict(
        nU=u.size[0], nG=nG, nF=nF,
        gradTolerance=float(gradTolerance), addEye2HessianUtolerance=float(addEye2HessianUtolerance),
        equalTolerance=float(equalTolerance), desiredDualityGap=float(desiredDualityGap),
        scaleCost=int(scaleCost != 0), scaleInequalities=int(bool(scaleInequalities)),
        scaleEqualities=int(bool(scaleEqualities)), alphaMin=float(alphaMin), alphaMax=float(alphaMax),
        coupledAlphas=int(bool(coupledAlphas)), setAddEye2Hessian=int(bool(addEye2Hessian)),
        adjustAddEye2Hessian=int(bool(adjustAddEye2Hessian)), useInertia=int(bool(useInertia)),
        muFactorAggressive=float(muFactorAggressive), muFactorConservative=float(muFactorConservative),
        delta=int(delta), skipAffine=int(bool(skipAffine) or nF == 0), smallerNewtonMatrix=int(bool(smallerNewtonMatrix)), useLDL=1, useUmfpack=0, allowSave=0,
        debugConvergence=0, profiling=0, verboseLevel=int(solverVerboseLevel))
    tape = code.compile(defines)
    if allowSave:
        import os
        from . import typical
        classdir = os.path.join(folder, "@" + classname)
        os.makedirs(classdir, exist_ok=True)
        wsubs, _ = code.scalarize(Tout["Hess"])
        typical.write_subscripts(os.path.join(classdir, classname + "_WW.subscripts"), Tout["Hess"].size, wsubs,
                                 typical.new_magic())
    spec = SolverSpec(classname, tape, defines,
                      [(p.par, p.size) for p in parameters],
                      [(v.par, v.size) for v in optimizationVariables],
                      [(nm, o.size) for nm, o in zip(outputNames, outputExpressions)], folder)
    spec.code = code
    spec.Tout = Tout
    return spec
