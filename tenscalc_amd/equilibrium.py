"""Two-player equilibrium solvers with latent variables: ``cmex2equilibriumLatentCS`` with a HIP back-end.

Restates /root/reference/lib/cmex2equilibriumLatentCS.m (front door: argument checks :58-154, sets :190-262,
constraint parsing and variable packing :266-369, the ``#define`` block :417-448) and
/root/reference/lib/ipmPDeqlat_CS.m (symbolic assembly: both Newton-matrix branches, the Mehrotra predictor step,
LU or symmetrised LDL).  Player 1 minimises ``P1objective`` over ``u``, player 2 minimises ``P2objective`` over ``d``;
latent variables ``x`` are defined by equality constraints ``H(u,d,x)=0`` both players see.  The first-order
conditions of the two problems are solved TOGETHER by a primal-dual interior-point method whose Newton matrix is not
symmetric -- hence the symbolic LU of csparse._s_lu (@csparse/sparsity_lu.m) -- and whose loop is the reference's
lib/ipmPDeq_CSsolver.c (device restatement: csrc/ipm_newton_eq.hip.inc).
"""
from __future__ import annotations

import re

from . import tcalculus as tc
from .cmex2optimizeCS import SolverSpec, packExpressions, packVariables
from .csparse import CSparse
from .tcalculus import (Teye, Tones, Tvariable, Tzeros, cat, clp, diag, full, gradient, gradientVector, ldl, lu, mldivide,
                        norminf, rdivide, substitute, tabs, times, tmin, tprod, transpose)

__all__ = ["cmex2equilibriumLatentCS", "ipmPDeqlat_CS"]


def _n(v):
    return v.size[0] if v.ndim else 0


def _vcat(*xs):
    xs = [x for x in xs if x is not None and _n(x) > 0]
    if not xs:
        return Tzeros((0,))
    return xs[0] if len(xs) == 1 else cat(1, *xs)


def ipmPDeqlat_CS(code, *, u, d, x, f, g, Fu, Fd, Gu, Gd, H, P1lambda, P1nu, P1xnu, P2lambda, P2nu, P2xnu,
                  addEye2Hessian=False, skipAffine=True, smallerNewtonMatrix=False, scaleCost=0.0,
                  scaleInequalities=True, useLDL=False, classname="cls"):
    """lib/ipmPDeqlat_CS.m: registers the tape entry points lib/ipmPDeq_CSsolver.c calls (:28-51) on ``code`` and
    returns the named expressions (``Hess``, ``Lf_ux`` ...)."""
    out = {}
    nU, nD, nX = _n(u), _n(d), _n(x)
    nZ = nU + nD + nX
    nGu, nGd, nH = _n(Gu), _n(Gd), _n(H)
    nG = nGu + nGd + nH
    nFu, nFd = _n(Fu), _n(Fd)
    nF = nFu + nFd

    # -- scaling (:78-106) --------------------------------------------------------------------------------
    if scaleInequalities:
        src, dst = [], []
        if nFu > 0:
            src.append(tabs(rdivide(1.0, Fu)))
            dst.append(Tvariable("scale4IneqU__", Fu.size))
            Fu = times(dst[-1], Fu)
        if nFd > 0:
            src.append(tabs(rdivide(1.0, Fd)))
            dst.append(Tvariable("scale4IneqD__", Fd.size))
            Fd = times(dst[-1], Fd)
        if src:
            code.declareCopy(dst, src, "scaleIneq__")
    if scaleCost > 0:
        scale4Cost = Tvariable("scale4Cost__", ())
        code.declareCopy(scale4Cost, tabs(rdivide(float(scaleCost), f)), "scaleCost__")
        f = scale4Cost * f
        g = scale4Cost * g
        code.declareGet(scale4Cost, "getScale4Cost__")
    code.declareGet([f, g], "getfg__")

    # -- stacked variables (:112-133): G = [Gu; Gd; H], nu = [P1nu; P1xnu; P2nu; P2xnu] ---------------------
    F = _vcat(Fu, Fd)
    lam = _vcat(P1lambda, P2lambda)
    G = _vcat(Gu, Gd, H)
    nu = _vcat(P1nu, P1xnu, P2nu, P2xnu)
    gap = tprod(lam, (-1,), F, (-1,)) if nF > 0 else Tzeros(())

    # -- exit conditions (:140-151) ---------------------------------------------------------------------
    if nF > 0:
        mu = Tvariable("mu__", ())
        muOnes = tc.expand(mu, (nF,))
        code.declareSet(mu, "setMu__")
        code.declareGet([gap, tmin(F, 1), tmin(lam, 1)], "getGapMinFMinLambda__")
    if nG > 0:
        code.declareGet(norminf(G), "getNorminf_G__")

    # -- initialisation of the duals (:153-190) -----------------------------------------------------------
    dst, src = [], []
    if nGu > 0:
        dst.append(P1nu); src.append(Tones((nGu,)))
    if nGd > 0:
        dst.append(P2nu); src.append(Tones((nGd,)))
    if nH > 0:
        dst += [P1xnu, P2xnu]; src += [Tones((nH,)), Tones((nH,))]
    if dst:
        code.declareCopy(dst, src, "initDualEq__")
    if nF > 0:
        dst, src = [], []
        if nFu > 0:
            dst.append(P1lambda); src.append(rdivide(tc.expand(mu, (nFu,)), Fu))
        if nFd > 0:
            dst.append(P2lambda); src.append(rdivide(tc.expand(mu, (nFd,)), Fd))
        code.declareCopy(dst, src, "initDualIneq__")

    # -- the two Lagrangians (:192-214) -------------------------------------------------------------------
    Lf, Lg = f, g
    if nFu > 0:
        Lf = Lf - tprod(P1lambda, (-1,), Fu, (-1,))
    if nFd > 0:
        Lg = Lg - tprod(P2lambda, (-1,), Fd, (-1,))
    if nGu > 0:
        Lf = Lf + tprod(P1nu, (-1,), Gu, (-1,))
    if nGd > 0:
        Lg = Lg + tprod(P2nu, (-1,), Gd, (-1,))
    if nH > 0:
        Lf = Lf + tprod(P1xnu, (-1,), H, (-1,))
        Lg = Lg + tprod(P2xnu, (-1,), H, (-1,))

    alphaPrimal = Tvariable("alphaPrimal__", ())
    code.declareSet(alphaPrimal, "setAlphaPrimal__")
    alphaDualEq = Tvariable("alphaDualEq__", ())
    code.declareSet(alphaDualEq, "setAlphaDualEq__")
    alphaDualIneq = Tvariable("alphaDualIneq__", ())
    code.declareSet(alphaDualIneq, "setAlphaDualIneq__")
    if addEye2Hessian:
        addEye2Hessian1 = Tvariable("addEye2Hessian1__", ())
        addEye2Hessian2 = Tvariable("addEye2Hessian2__", ())
        code.declareSet(addEye2Hessian1, "setAddEye2Hessian1__")
        code.declareSet(addEye2Hessian2, "setAddEye2Hessian2__")
    else:
        addEye2Hessian1 = Tzeros(())
        addEye2Hessian2 = Tzeros(())
    out["addEye2Hessian1"], out["addEye2Hessian2"] = addEye2Hessian1, addEye2Hessian2

    # -- first derivatives and the gradient norm of the exit test (:236-253) ----------------------------------
    Lf_u = gradient(Lf, u)
    Lg_d = gradient(Lg, d)
    if nX > 0:
        Lf_x = gradient(Lf, x)
        Lg_x = gradient(Lg, x)
        code.declareGet(norminf(Lf_u) + norminf(Lf_x) + norminf(Lg_d) + norminf(Lg_x), "getNorminf_Grad__")
    else:
        Lf_x = Lg_x = Tzeros((0,))
        code.declareGet(norminf(Lf_u) + norminf(Lg_d), "getNorminf_Grad__")

    # -- second derivatives (:258-293) -----------------------------------------------------------------------
    zs = [u, d, x] if nX > 0 else [u, d]
    ls = [P1lambda, P2lambda]
    ns = [P1nu, P1xnu, P2nu, P2xnu]
    Lf_uz, Lg_dz = gradientVector(Lf_u, zs), gradientVector(Lg_d, zs)
    Lf_ul, Lg_dl = gradientVector(Lf_u, ls), gradientVector(Lg_d, ls)
    Lf_un, Lg_dn = gradientVector(Lf_u, ns), gradientVector(Lg_d, ns)
    if nX > 0:
        Lf_xz, Lg_xz = gradientVector(Lf_x, zs), gradientVector(Lg_x, zs)
        Lf_xl, Lg_xl = gradientVector(Lf_x, ls), gradientVector(Lg_x, ls)
        Lf_xn, Lg_xn = gradientVector(Lf_x, ns), gradientVector(Lg_x, ns)
    F_z = gradientVector(F, zs) if nF > 0 else Tzeros((0, nZ))
    G_z = gradientVector(G, zs) if nG > 0 else Tzeros((0, nZ))
    # derivatives to confirm minimum / maximum (:295-305)
    if nX > 0:
        out["Lf_ux"] = cat(1, Lf_u, Lf_x)
        out["Lf_uxux"] = gradientVector(out["Lf_ux"], [u, x])
        out["Lg_dx"] = cat(1, Lg_d, Lg_x)
        out["Lg_dxdx"] = gradientVector(out["Lg_dx"], [d, x])
    else:
        out["Lf_ux"], out["Lf_uxux"] = Lf_u, gradient(Lf_u, u)
        out["Lg_dx"], out["Lg_dxdx"] = Lg_d, gradient(Lg_d, d)
    if nF > 0:
        LFF = tprod(rdivide(lam, F), (1,), F_z, (1, 2))           # diag(lambda./F)*F_z  (:313)
        LFF = code.declareAlias(LFF, "LFF__")

    def hcat(*blocks):
        blocks = [b for b in blocks if b.size[1] > 0]
        return blocks[0] if len(blocks) == 1 else cat(2, *blocks)

    def vstack(rows):
        rows = [r for r in rows if r.size[0] > 0]
        return rows[0] if len(rows) == 1 else cat(1, *rows)

    nNu = nG + nH
    if smallerNewtonMatrix:
        # -- small matrix (:323-431): the inequality duals eliminated -----------------------------------------
        def red(Lzz, Lzl):
            return Lzz - Lzl * LFF if nF > 0 else Lzz
        if nX > 0:
            WW = vstack([hcat(red(Lf_uz, Lf_ul), Lf_un), hcat(red(Lg_dz, Lg_dl), Lg_dn),
                         hcat(red(Lf_xz, Lf_xl), Lf_xn), hcat(red(Lg_xz, Lg_xl), Lg_xn),
                         hcat(G_z, Tzeros((nG, nNu)))])
            out["Hess"] = WW
        else:
            top = vstack([red(Lf_uz, Lf_ul), red(Lg_dz, Lg_dl)])
            WW = vstack([hcat(top + addEye2Hessian1 * Teye((nZ, nZ)), vstack([Lf_un, Lg_dn])),
                         hcat(G_z, -(addEye2Hessian2 * Teye((nG, nG))))] if nG > 0 else
                        [top + addEye2Hessian1 * Teye((nZ, nZ))])
            out["Hess"] = vstack([hcat(top, vstack([Lf_un, Lg_dn])), hcat(G_z, Tzeros((nG, nG)))] if nG > 0 else [top])
        ff, gg = f, g
        if nGu > 0:
            ff = ff + tprod(P1nu, (-1,), Gu, (-1,))
        if nGd > 0:
            gg = gg + tprod(P2nu, (-1,), Gd, (-1,))
        if nH > 0 and nX > 0:
            ff = ff + tprod(P1xnu, (-1,), H, (-1,))
            gg = gg + tprod(P2xnu, (-1,), H, (-1,))
        parts = [gradient(ff, u), gradient(gg, d)] + ([gradient(ff, x), gradient(gg, x)] if nX > 0 else [])
        b_a = -_vcat(*(parts + [G]))
        b_s = b_a
        Ls = [Lf_ul, Lg_dl] + ([Lf_xl, Lg_xl] if nX > 0 else [])
        if nF > 0:
            b_s = b_s - _vcat(*([L_ * rdivide(muOnes, F) for L_ in Ls] + [Tzeros((nG,))]))
    else:
        # -- large matrix (:559-622) ----------------------------------------------------------------------------
        rows = [hcat(Lf_uz, Lf_un, Lf_ul), hcat(Lg_dz, Lg_dn, Lg_dl)]
        if nX > 0:
            rows += [hcat(Lf_xz, Lf_xn, Lf_xl), hcat(Lg_xz, Lg_xn, Lg_xl)]
        if nG > 0:
            rows.append(hcat(G_z, Tzeros((nG, nNu + nF))))
        if nF > 0:
            rows.append(hcat(F_z, Tzeros((nF, nNu)), diag(rdivide(F, lam))))
        WW = vstack(rows)
        out["Hess"] = WW
        grads = [-Lf_u, -Lg_d] + ([-Lf_x, -Lg_x] if nX > 0 else [])
        b_a = _vcat(*(grads + ([-G] if nG > 0 else []) + ([-F] if nF > 0 else [])))
        b_s = _vcat(*(grads + ([-G] if nG > 0 else []) + ([-F + rdivide(muOnes, lam)] if nF > 0 else [])))

    WWraw = code.declareAlias(WW, "WW__")
    WW = WWraw
    if useLDL:        # symmetrise (:430-436, :645-651)
        b_s = transpose(WW) * b_s
        b_a = transpose(WW) * b_a
        WW = transpose(WW) * WW
        lu_ww = ldl(WW, classname + "_WW")
    else:
        lu_ww = lu(WW, classname + "_WW")

    dZ_a = None
    if not skipAffine:
        # -- affine scaling direction (:443-497, :659-714) -----------------------------------------------------
        dx_a = code.declareAlias(mldivide(lu_ww, b_a), "dZNu_a__" if smallerNewtonMatrix else "dx_a__")
        dU_a, dD_a, dZ_a = dx_a[0:nU], dx_a[nU:nU + nD], dx_a[0:nZ]
        newU_a = u + alphaPrimal * dU_a
        newD_a = d + alphaPrimal * dD_a
        if nX > 0:
            newX_a = x + alphaPrimal * dx_a[nU + nD:nZ]
        if nF > 0:
            if smallerNewtonMatrix:
                dLambda_a = -lam - LFF * dZ_a
            else:
                dLambda_a = dx_a[nZ + nNu:nZ + nNu + nF]
            newLambda_a = lam + alphaDualIneq * dLambda_a
            newF_a = substitute(substitute(F, u, newU_a), d, newD_a)
            if nX > 0:
                newF_a = substitute(newF_a, x, newX_a)
            maxAlphaPrimal_a = clp(F, F_z * dZ_a)
            maxAlphaDualIneq_a = clp(lam, dLambda_a)
            if smallerNewtonMatrix:
                Mehrotra = rdivide(times(F_z * dZ_a, dLambda_a), F)
                b_s = b_s + _vcat(*([L_ * Mehrotra for L_ in Ls] + [Tzeros((nG,))]))
            else:
                Mehrotra = rdivide(times(F_z * dZ_a, dLambda_a), lam)
                b_s = _vcat(*(grads + ([-G] if nG > 0 else []) + [-F - Mehrotra + rdivide(muOnes, lam)]))
                if useLDL:
                    # DELIBERATE DEVIATION from ipmPDeqlat_CS.m:702-707: the reference rebuilds b_s here WITHOUT the WW'
                    # factor it applied at :646-648, i.e. it solves (WW'WW) dx = b_s with an un-symmetrised right-hand side
                    # whenever skipAffine=false meets useLDL=true.  The small-matrix branch above follows the reference's
                    # formula to the letter (:419-421 then :473-479, the same slip) and, like the reference's solver, then
                    # walks into maxIter; this branch keeps the product, so the option pair is usable with the default
                    # (large) matrix.  tests/test_equilibrium.py::test_predictor_step_on_the_symmetrised_system, DESIGN.md 7.
                    b_s = transpose(WWraw) * b_s
            rho = rdivide(tprod(newF_a, (-1,), newLambda_a, (-1,)), gap)
            code.declareGet([maxAlphaPrimal_a, maxAlphaDualIneq_a], "getMaxAlphas_a__")
            code.declareGet(tmin(newF_a, 1), "getMinF_a__")
            code.declareGet(rho, "getRho__")

    # -- combined search direction (:499-556, :716-766) ---------------------------------------------------------
    dx_s = code.declareAlias(mldivide(lu_ww, b_s), "dZNu_s__" if smallerNewtonMatrix else "dx_s__")
    code.declareGet([norminf(WW * dx_s - b_s)], "getDirectionError__")
    dU_s, dD_s, dZ_s = dx_s[0:nU], dx_s[nU:nU + nD], dx_s[0:nZ]
    o = nZ
    dNuU_s = dx_s[o:o + nGu]
    dNuUx_s = dx_s[o + nGu:o + nGu + nH]
    dNuD_s = dx_s[o + nGu + nH:o + nGu + nH + nGd]
    dNuDx_s = dx_s[o + nGu + nH + nGd:o + nGu + nH + nGd + nH]
    if nF > 0:
        if smallerNewtonMatrix:
            dLambda_s = -lam - LFF * dZ_s + rdivide(muOnes, F)
            if not skipAffine:
                dLambda_s = dLambda_s - Mehrotra
        else:
            dLambda_s = dx_s[nZ + nNu:nZ + nNu + nF]
    newU_s = u + alphaPrimal * dU_s
    newD_s = d + alphaPrimal * dD_s
    if nX > 0:
        newX_s = x + alphaPrimal * dx_s[nU + nD:nZ]
    if nF > 0:
        maxAlphaPrimal_s = clp(F, F_z * dZ_s)
        maxAlphaDualIneq_s = clp(lam, dLambda_s)
        newLambda_s = lam + alphaDualIneq * dLambda_s
        newF_s = substitute(substitute(F, u, newU_s), d, newD_s)
        if nX > 0:
            newF_s = substitute(newF_s, x, newX_s)
        code.declareGet([maxAlphaPrimal_s, maxAlphaDualIneq_s], "getMaxAlphas_s__")
        code.declareGet(tmin(newF_s, 1), "getMinF_s__")

    # -- updatePrimalDual__ (:804-836) ----------------------------------------------------------------------
    dst, src = [u, d], [newU_s, newD_s]
    if nX > 0:
        dst.append(x); src.append(newX_s)
    if nGu > 0:
        dst.append(P1nu); src.append(P1nu + alphaDualEq * dNuU_s)
    if nGd > 0:
        dst.append(P2nu); src.append(P2nu + alphaDualEq * dNuD_s)
    if nH > 0:
        dst += [P1xnu, P2xnu]
        src += [P1xnu + alphaDualEq * dNuUx_s, P2xnu + alphaDualEq * dNuDx_s]
    if nFu > 0:
        dst.append(P1lambda); src.append(newLambda_s[0:nFu])
    if nFd > 0:
        dst.append(P2lambda); src.append(newLambda_s[nFu:nF])
    code.declareCopy(dst, src, "updatePrimalDual__")
    out.update(u=u, d=d, x=x, F=F, G=G, nu=nu, dx_s=dx_s, b_s=b_s, WW=WWraw)
    out["lambda"] = lam
    return out


def _parse(code, classname, constraints, prefix, outputExpressions):
    """lib/private/parseConstraints.m with a dual-variable prefix ('P1', 'P2', 'P1x', 'P2x'): equality constraints ->
    G and <prefix>nu<k>_ duals, inequalities -> F and <prefix>lambda<k>_ duals"""
    Gc, Fc, nus, lambdas = [], [], [], []
    for k, c in enumerate(constraints):
        if not isinstance(c, tc.Tcalculus):
            raise ValueError(f"constraint {k + 1} is not a Tcalculus expression")
        op1 = c.args[0]
        if c.op == "iszero":
            Gc.append(op1)
            v = Tvariable(f"{prefix}nu{len(nus) + 1}_", op1.size)
            nus.append(v)
        elif c.op == "ispositive":
            Fc.append(op1)
            v = Tvariable(f"{prefix}lambda{len(lambdas) + 1}_", op1.size)
            lambdas.append(v)
        else:
            raise ValueError(f"constraint of type '{c.op}' not implemented")
        code.declareSet(v, f"{classname}_set_{v.par}")
    return packExpressions(Gc), packExpressions(Fc), nus, lambdas


def cmex2equilibriumLatentCS(*, classname, P1objective, P2objective, P1optimizationVariables, P2optimizationVariables,
                             latentVariables=(), P1constraints=(), P2constraints=(), latentConstraints=(),
                             parameters=(), outputExpressions=(), outputNames=None, folder=".",
                             gradTolerance=1e-4, equalTolerance=1e-4, desiredDualityGap=1e-5,
                             scaleInequalities=True, scaleCost=0.0, scaleEqualities=False,
                             muFactorAggressive=0.2, muFactorConservative=0.95, skipAffine=True, delta=3,
                             alphaMin=1e-7, alphaMax=1.0, coupledAlphas=True, addEye2Hessian=False,
                             adjustAddEye2Hessian=False, addEye2Hessian1tolerance=1e-6, smallerNewtonMatrix=False,
                             useLDL=False, useUmfpack=False, allowSave=False, solverVerboseLevel=0, **ignored):
    """Generate a solver for the two-player equilibrium

        u* = argmin_u { P1objective(u, d*, x) : P1constraints, latentConstraints }
        d* = argmin_d { P2objective(u*, d, x) : P2constraints, latentConstraints }

    (cmex2equilibriumLatentCS.m; option defaults from lib/private/parameters4equilibrium.m:97-182 and
    parameters4all.m).  Returns a :class:`~tenscalc_amd.cmex2optimizeCS.SolverSpec` whose ``solver`` is
    ``"ipmPDeq_CSsolver"``: the class keeps the reference surface ``setP_*/setV_*/solve/getOutputs``."""
    if ignored:
        accepted_noop = {"compilerOptimization", "compileGateways", "compileLibrary", "compileStandalones", "callType",
                         "absolutePath", "profiling", "umfpack", "debugConvergence", "debugConvergenceThreshold",
                         "scratchbookType", "fastRedundancyCheck", "codeType", "minInstructions4loop",
                         "maxInstructionsPerFunction", "LDLthreshold"}
        bad = sorted(set(ignored) - accepted_noop)
        if bad:
            raise TypeError("cmex2equilibriumLatentCS: unknown option(s) %s" % ", ".join(bad))
    if useUmfpack:
        raise NotImplementedError("useUmfpack is not supported by the HIP back-end")
    classname = re.sub(r"[-+]", "_", classname)
    P1objective, P2objective = tc.toT(P1objective), tc.toT(P2objective)
    if P1objective.size != ():
        raise ValueError(f"P1's minimization criterion must be scalar (not {list(P1objective.size)})")
    if P2objective.size != ():
        raise ValueError(f"P2's minimization criterion must be scalar (not {list(P2objective.size)})")
    P1v, P2v, Lv = list(P1optimizationVariables), list(P2optimizationVariables), list(latentVariables)
    parameters = list(parameters)
    for who, vs in (("P1optimizationVariables", P1v), ("P2optimizationVariables", P2v), ("latentVariables", Lv)):
        for k, v in enumerate(vs):
            if not isinstance(v, tc.Tcalculus) or v.op != "variable":
                raise ValueError(f"all {who} must be of the type 'variable' ({k + 1}th is not)")
            for p in parameters:
                if p.par == v.par:
                    raise ValueError(f"optimization variable '{v.par}' cannot also be a parameter")
    outputExpressions = [tc.toT(o) for o in outputExpressions]
    if outputNames is None:
        outputNames = [o.par if o.op == "variable" else f"y{k + 1}" for k, o in enumerate(outputExpressions)]

    code = CSparse()
    import os
    classdir = os.path.join(folder, "@" + classname)
    code.typical_folder = classdir     # <cls>_WW.subscripts/.values of an earlier run order the LU (sparsity_lu.m:36-57)
    for p in parameters:
        code.declareSet(p, f"{classname}_set_{p.par}", external=True)
    for v in P1v + P2v + Lv:
        code.declareSet(v, f"{classname}_set_{v.par}", external=True)
    Gu, Fu, P1nus, P1lambdas = _parse(code, classname, list(P1constraints), "P1", outputExpressions)
    Gd, Fd, P2nus, P2lambdas = _parse(code, classname, list(P2constraints), "P2", outputExpressions)
    H, Fh, P1xnus, _ = _parse(code, classname, list(latentConstraints), "P1x", outputExpressions)
    _, _, P2xnus, _ = _parse(code, classname, list(latentConstraints), "P2x", outputExpressions)
    if _n(Fh) > 0:
        raise ValueError("latent constraints cannot be inequalities")

    def pack(vs, name, exprs):
        if not vs:
            return Tzeros((0,)), exprs
        r = packVariables(vs, name, *exprs)
        return r[0], list(r[2:])
    exprs = [P1objective, P2objective, Fu, Gu, Fd, Gd, H] + outputExpressions
    u, exprs = pack(P1v, "u_", exprs)
    d, exprs = pack(P2v, "d_", exprs)
    x, exprs = pack(Lv, "x_", exprs)
    if _n(H) != _n(x):
        raise ValueError("number of latent variables (=%d) must match number of latent (equality) constraints (=%d) so "
                         "that latent variables are uniquely defined" % (_n(x), _n(H)))
    src = [packExpressions(P1v), packExpressions(P2v)] + ([packExpressions(Lv)] if Lv else [])
    dst = [u, d] + ([x] if Lv else [])
    code.declareCopy(dst, src, "initPrimal__")
    duals = {}
    for key, vs, n_ in (("P1nu", P1nus, _n(exprs[3])), ("P1lambda", P1lambdas, _n(exprs[2])), ("P2nu", P2nus, _n(exprs[5])),
                        ("P2lambda", P2lambdas, _n(exprs[4])), ("P1xnu", P1xnus, _n(exprs[6])), ("P2xnu", P2xnus, _n(exprs[6]))):
        if n_ > 0:
            duals[key], exprs = pack(vs, key + "_", exprs)
        else:
            duals[key] = Tzeros((0,))
    P1objective, P2objective, Fu, Gu, Fd, Gd, H = exprs[:7]
    outputExpressions = list(exprs[7:])

    Tout = ipmPDeqlat_CS(code, u=u, d=d, x=x, f=P1objective, g=P2objective, Fu=Fu, Fd=Fd, Gu=Gu, Gd=Gd, H=H,
                         P1lambda=duals["P1lambda"], P1nu=duals["P1nu"], P1xnu=duals["P1xnu"],
                         P2lambda=duals["P2lambda"], P2nu=duals["P2nu"], P2xnu=duals["P2xnu"],
                         addEye2Hessian=bool(addEye2Hessian), skipAffine=bool(skipAffine),
                         smallerNewtonMatrix=bool(smallerNewtonMatrix), scaleCost=scaleCost,
                         scaleInequalities=bool(scaleInequalities), useLDL=bool(useLDL), classname=classname)
    for fn in ("Hess", "Lf_ux", "Lf_uxux", "Lg_dx", "Lg_dxdx"):     # cmex2equilibriumLatentCS.m:385-394
        val = Tout[fn]
        ph = Tvariable(fn + "_", val.size)
        outputExpressions = [substitute(o, ph, val) for o in outputExpressions]
    outputExpressions = [full(o) for o in outputExpressions]
    code.declareGet(outputExpressions, f"{classname}_getOutputs", list(outputNames))
    if allowSave:        # ipmPDeqlat_CS.m:838-850: declareSave(code,WW,'saveWW__',<cls>_WW.subscripts)
        code.declareGet([full(Tout["WW"])], "saveWW__", ["WW"])

    pe = lambda v: float("%e" % float(v))          # noqa: E731  (sprintf('%e'), cmex2equilibriumLatentCS.m:425-438)
    nU, nD, nX = _n(u), _n(d), _n(x)
    nG = _n(Gu) + _n(Gd) + _n(H)
    nF = _n(Fu) + _n(Fd)
    defines = dict(
        nZ=nU + nD + nX, nU=nU, nD=nD, nX=nX, nG=nG, nF=nF, nNu=_n(Gu) + _n(Gd) + 2 * _n(H),
        gradTolerance=pe(gradTolerance), addEye2Hessian1tolerance=pe(addEye2Hessian1tolerance),
        equalTolerance=pe(equalTolerance), desiredDualityGap=pe(desiredDualityGap),
        scaleCost=int(scaleCost != 0), scaleInequalities=int(bool(scaleInequalities)),
        scaleEqualities=int(bool(scaleEqualities)), alphaMin=pe(alphaMin), alphaMax=pe(alphaMax),
        coupledAlphas=int(bool(coupledAlphas)), setAddEye2Hessian=int(bool(addEye2Hessian)),
        adjustAddEye2Hessian=int(bool(adjustAddEye2Hessian)), muFactorAggressive=pe(muFactorAggressive),
        muFactorConservative=pe(muFactorConservative), delta=int(delta), skipAffine=int(bool(skipAffine) or nF == 0),
        smallerNewtonMatrix=int(bool(smallerNewtonMatrix)), useLDL=int(bool(useLDL)), useUmfpack=0, allowSave=0,      # (the LOOP never saves: HipSolver.saveWW does)
        debugConvergence=0, debugConvergenceThreshold=1e5, profiling=0, verboseLevel=int(solverVerboseLevel))
    tape = code.compile(defines)
    if allowSave:        # after compile: the LU above has used the pair an EARLIER run left (ipmPDeqlat_CS.m:836-837)
        from . import typical
        os.makedirs(classdir, exist_ok=True)
        wsubs, _ = code.scalarize(Tout["WW"])
        typical.write_subscripts(os.path.join(classdir, classname + "_WW.subscripts"), Tout["WW"].size, wsubs,
                                 typical.new_magic())
    spec = SolverSpec(classname, tape, defines, [(p.par, p.size) for p in parameters],
                      [(v.par, v.size) for v in P1v + P2v + Lv],
                      [(nm, o.size) for nm, o in zip(outputNames, outputExpressions)], folder)
    spec.solver = "ipmPDeq_CSsolver"
    spec.code = code
    spec.Tout = Tout
    return spec
