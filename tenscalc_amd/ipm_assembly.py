"""Symbolic assembly of the primal-dual interior-point step (code-generation time).

Restates /root/reference/lib/ipmPD_CStimesLambda.m (the variant ``cmex2optimizeCS``
hard-selects, cmex2optimizeCS.m:83-95) for ``packOptimizationVariables=true``:
Lagrangian, gradient, Hessian, KKT matrix ``WW``, right-hand side ``b``,
``dx=ldl(WW)\\b`` and the multiplicative-lambda update, registered on a
:class:`~tenscalc_amd.csparse.CSparse` object under the entry-point names the Newton
loop calls (``lib/ipmPD_CSsolver.c:28-51``).
"""
from __future__ import annotations

import numpy as np

from . import tcalculus as tc
from .tcalculus import (Teye, Tones, Tvariable, Tzeros, cat, clp, compose, diag, full,
                        gradient, hessian, ldl, ldl_d, ldl_l, mldivide, norminf,
                        rdivide, substitute, tabs, times, tmin, tprod, transpose, tsum)


def ipmPD_CStimesLambda(code, u, f, F, G, lam, nu, *, addEye2Hessian=True, scaleCost=0.0,
                        scaleInequalities=True, classname="cls", rewrite=None):
    """Register every tape entry point of the IPM on ``code``; returns the named
    expressions (``u, F, G, nu, lambda, Lf_u, Lf_uu, Hess, dHess, lHess, b, dx, mu``).
    ``rewrite``: optional expression rewriter applied to the derivatives before they are used
    (tenscalc_amd/hoist.py; not part of the reference)."""
    out = {}
    rw = rewrite if rewrite is not None else (lambda e: e)
    nU = u.size[0]
    nG = G.size[0] if G.ndim else 0
    nF = F.size[0] if F.ndim else 0

    # -- scaling (ipmPD_CStimesLambda.m:86-103) --------------------------------------
    if nF > 0 and scaleInequalities:
        scale4Ineq = Tvariable("scale4Ineq__", F.size)
        code.declareCopy(scale4Ineq, tabs(rdivide(1.0, F)), "scaleIneq__")
        F = times(scale4Ineq, F)
    if scaleCost > 0:
        scale4Cost = Tvariable("scale4Cost__", ())
        code.declareCopy(scale4Cost, tabs(rdivide(float(scaleCost), f)), "scaleCost__")
        f = scale4Cost * f
        code.declareGet(scale4Cost, "getScale4Cost__")
    out["u"], out["F"], out["G"], out["nu"], out["lambda"] = u, F, G, nu, lam

    # -- matrices used to modify the Hessian's inertia (:105-116) ------------------------
    if addEye2Hessian:
        addEye2HessianU = Tvariable("addEye2HessianU__", ())
        addEye2HessianEq = Tvariable("addEye2HessianEq__", ())
        code.declareSet(addEye2HessianU, "setAddEye2HessianU__")
        code.declareSet(addEye2HessianEq, "setAddEye2HessianEq__")
    else:
        addEye2HessianU = Tzeros(())
        addEye2HessianEq = Tzeros(())
    out["addEye2HessianU"], out["addEye2HessianEq"] = addEye2HessianU, addEye2HessianEq

    # -- cost (:121) -------------------------------------------------------------------------
    code.declareGet(rw(f), "getJ__")

    # -- Lagrangian and derivatives (:124-187) ---------------------------------------------------
    f_u = rw(gradient(f, u))
    Lf = f
    if nF > 0:
        F_u = rw(gradient(F, u))
        gap = tprod(lam, (-1,), F, (-1,))
        Lf = Lf - gap
    else:
        F = Tzeros((0,))
        F_u = Tzeros((0, nU))
        gap = Tzeros(())
    if nG > 0:
        G_u = rw(gradient(G, u))
        Lf = Lf + tprod(nu, (-1,), G, (-1,))
    else:
        G = Tzeros((0,))
        G_u = Tzeros((0, nU))
    Lf_uu, Lf_u = hessian(Lf, u)
    Lf_uu, Lf_u = rw(Lf_uu), rw(Lf_u)
    out["Lf_u"], out["Lf_uu"] = Lf_u, Lf_uu
    code.declareGet(Lf_u, "getGrad__")
    code.declareGet(norminf(Lf_u), "getNorminf_Grad__")

    # -- barrier variable (:189-205) ------------------------------------------------------------
    if nF > 0:
        mu = Tvariable("mu__", ())
        muOnes = tc.expand(mu, (nF,))
        code.declareSet(mu, "setMu__")
        code.declareCopy(lam, rdivide(muOnes, F), "initDualIneq__")
        code.declareGet([gap, tmin(F, 1), tmin(lam, 1)], "getGapMinFMinLambda__")
    else:
        mu = Tzeros(())
        muOnes = Tzeros((0,))
    out["mu"] = mu

    if nG > 0:
        # automatic initialisation of nu (:207-221): simple and least-squares versions
        code.declareCopy(nu, Tones((nG,)), "initDualEq__")
        WW0 = cat(1, cat(2, Teye((nU, nU)), transpose(G_u)),
                  cat(2, G_u, -(addEye2HessianEq * Teye((nG, nG)))))
        factor_ww0 = ldl(WW0, classname + "_WW0", stages=[0] * nU + [1] * nG)
        b0 = cat(1, (transpose(F_u) * lam if nF > 0 else Tzeros((nU,))) - f_u, Tzeros((nG,)))
        wnu0 = full(mldivide(factor_ww0, b0))
        code.declareCopy(nu, wnu0[nU:nU + nG], "initDualEqX__")
        code.declareGet(norminf(G), "getNorminf_G__")

    # -- step sizes (:223-229) -----------------------------------------------------------------------
    alphaPrimal = Tvariable("alphaPrimal__", ())
    code.declareSet(alphaPrimal, "setAlphaPrimal__")
    alphaDualEq = Tvariable("alphaDualEq__", ())
    code.declareSet(alphaDualEq, "setAlphaDualEq__")
    alphaDualIneq = Tvariable("alphaDualIneq__", ())
    code.declareSet(alphaDualIneq, "setAlphaDualIneq__")

    # -- Newton step (:234-258) --------------------------------------------------------------------
    WW11 = Lf_uu + addEye2HessianU * Teye((nU, nU))
    rows = [[WW11], [], []]
    if nG > 0:
        rows[0].append(transpose(G_u))
        rows[1] = [G_u, -(addEye2HessianEq * Teye((nG, nG)))]
    if nF > 0:
        rows[0].append(-(transpose(F_u) * diag(lam)))
        if nG > 0:
            rows[1].append(Tzeros((nG, nF)))
        rows[2] = [-(diag(lam) * F_u)]
        if nG > 0:
            rows[2].append(Tzeros((nF, nG)))
        rows[2].append(-diag(times(F, lam)))
    WW = cat(1, *[cat(2, *r) for r in rows if r])
    out["Hess"] = WW
    # elimination order: fill-reducing within {primal, inequality duals}, equality duals last so
    # their -addEye2HessianEq pivots are never used before the primal block they couple
    factor_ww = ldl(WW, classname + "_WW", stages=[0] * nU + [1] * nG + [2] * nF)
    out["dHess"] = ldl_d(factor_ww)
    out["lHess"] = ldl_l(factor_ww)
    code.declareGet([tsum(compose(out["dHess"], "heaviside")),
                     tsum(compose(-out["dHess"], "heaviside"))], "getHessInertia__")

    bparts = [-Lf_u]
    if nG > 0:
        bparts.append(-G)
    if nF > 0:
        bparts.append(times(lam, F) - muOnes)
    b = cat(1, *bparts)
    dx = mldivide(factor_ww, b)
    dx = code.declareAlias(dx, "dx__")
    code.declareGet([norminf(WW * dx - b)], "getDirectionError__")

    # -- updates (:260-281) ---------------------------------------------------------------------------
    dU = dx[0:nU]
    newU = u + alphaPrimal * dU
    if nG > 0:
        dNu = dx[nU:nU + nG]
        newNu = nu + alphaDualEq * dNu
    curvature = dU * (WW11 * dU)
    code.declareGet(curvature, "getCurvature__")
    dsts, srcs = [u], [newU]
    if nG > 0:
        dsts.append(nu)
        srcs.append(newNu)
    if nF > 0:
        dLambda = dx[nU + nG:nU + nG + nF]
        newLambda = times(lam, Tones((nF,)) + alphaDualIneq * dLambda)
        maxAlphaPrimal = clp(F, F_u * dU)
        maxAlphaDualIneq = clp(Tones((nF,)), dLambda)
        code.declareGet([maxAlphaPrimal, maxAlphaDualIneq], "getMaxAlphas_s__")
        newF = substitute(F, u, newU)
        code.declareGet(tmin(newF, 1), "getMinF_s__")
        dsts.append(lam)
        srcs.append(newLambda)
    code.declareCopy(dsts, srcs, "updatePrimalDual__")
    out["b"], out["dx"] = b, dx
    return out


def ipmPD_CS(code, u, f, F, G, lam, nu, *, addEye2Hessian=True, scaleCost=0.0, scaleInequalities=True,
             classname="cls", skipAffine=True, smallerNewtonMatrix=False, rewrite=None, isSensitivity=None):
    """The reference's ORIGINAL assembly, /root/reference/lib/ipmPD_CS.m (packOptimizationVariables=true, LDL, no
    trust region): ADDITIVE update of lambda, optional Mehrotra predictor ("affine") step -- skipAffine=False:
    entry points getMaxAlphas_a__, getMinF_a__, getRho__ for lib/ipmPD_CSsolver.c:576-665 -- and optional
    elimination of the inequality duals from the Newton system (smallerNewtonMatrix, :249-333: a (nU+nG) square
    system instead of nU+nG+nF).  cmex2optimizeCS.m:83-95 hard-selects ipmPD_CStimesLambda (``switch 3``); this is
    what ``switch 1`` there generates."""
    out = {}
    rw = rewrite if rewrite is not None else (lambda e: e)
    nU = u.size[0]
    nG = G.size[0] if G.ndim else 0
    nF = F.size[0] if F.ndim else 0
    if nF > 0 and scaleInequalities:                                        # :90-97
        scale4Ineq = Tvariable("scale4Ineq__", F.size)
        code.declareCopy(scale4Ineq, tabs(rdivide(1.0, F)), "scaleIneq__")
        F = times(scale4Ineq, F)
    if scaleCost > 0:                                                       # :98-107
        scale4Cost = Tvariable("scale4Cost__", ())
        code.declareCopy(scale4Cost, tabs(rdivide(float(scaleCost), f)), "scaleCost__")
        f = scale4Cost * f
        code.declareGet(scale4Cost, "getScale4Cost__")
    out["u"], out["F"], out["G"], out["nu"], out["lambda"] = u, F, G, nu, lam
    code.declareGet(rw(f), "getJ__")
    if addEye2Hessian:                                                      # :128-137
        addEye2HessianU = Tvariable("addEye2HessianU__", ())
        addEye2HessianEq = Tvariable("addEye2HessianEq__", ())
        code.declareSet(addEye2HessianU, "setAddEye2HessianU__")
        code.declareSet(addEye2HessianEq, "setAddEye2HessianEq__")
    else:
        addEye2HessianU = Tzeros(())
        addEye2HessianEq = Tzeros(())
    f_u = rw(gradient(f, u))                                                # :146-200
    Lf = f
    if nF > 0:
        F_u = rw(gradient(F, u))
        gap = tprod(lam, (-1,), F, (-1,))
        Lf = Lf - gap
    else:
        F, F_u, gap = Tzeros((0,)), Tzeros((0, nU)), Tzeros(())
    if nG > 0:
        G_u = rw(gradient(G, u))
        Lf = Lf + tprod(nu, (-1,), G, (-1,))
    else:
        G, G_u = Tzeros((0,)), Tzeros((0, nU))
    Lf_uu, Lf_u = hessian(Lf, u)
    Lf_uu, Lf_u = rw(Lf_uu), rw(Lf_u)
    out["Lf_u"], out["Lf_uu"] = Lf_u, Lf_uu
    code.declareGet(Lf_u, "getGrad__")
    code.declareGet(norminf(Lf_u), "getNorminf_Grad__")
    if nF > 0:                                                              # :207-218
        mu = Tvariable("mu__", ())
        muOnes = tc.expand(mu, (nF,))
        code.declareSet(mu, "setMu__")
        code.declareCopy(lam, rdivide(muOnes, F), "initDualIneq__")
        code.declareGet([gap, tmin(F, 1), tmin(lam, 1)], "getGapMinFMinLambda__")
    else:
        mu, muOnes = Tzeros(()), Tzeros((0,))
    out["mu"] = mu
    if nG > 0:                                                              # :219-234
        code.declareCopy(nu, Tones((nG,)), "initDualEq__")
        WW0 = cat(1, cat(2, Teye((nU, nU)), transpose(G_u)),
                  cat(2, G_u, -(addEye2HessianEq * Teye((nG, nG)))))
        factor_ww0 = ldl(WW0, classname + "_WW0", stages=[0] * nU + [1] * nG)
        b0 = cat(1, (transpose(F_u) * lam if nF > 0 else Tzeros((nU,))) - f_u, Tzeros((nG,)))
        code.declareCopy(nu, full(mldivide(factor_ww0, b0))[nU:nU + nG], "initDualEqX__")
        code.declareGet(norminf(G), "getNorminf_G__")
    alphaPrimal = Tvariable("alphaPrimal__", ())
    code.declareSet(alphaPrimal, "setAlphaPrimal__")
    alphaDualEq = Tvariable("alphaDualEq__", ())
    code.declareSet(alphaDualEq, "setAlphaDualEq__")
    alphaDualIneq = Tvariable("alphaDualIneq__", ())
    code.declareSet(alphaDualIneq, "setAlphaDualIneq__")

    def blocks(rows):
        return cat(1, *[cat(2, *r) for r in rows if r])

    def affine(dU_a, dLambda_a):
        """entry points of the predictor step (:283-296 / :379-392)"""
        newU_a = u + alphaPrimal * dU_a
        newLambda_a = lam + alphaDualIneq * dLambda_a
        newF_a = substitute(F, u, newU_a)
        code.declareGet([clp(F, F_u * dU_a), clp(lam, dLambda_a)], "getMaxAlphas_a__")
        code.declareGet(tmin(newF_a, 1), "getMinF_a__")
        code.declareGet(rdivide(tprod(newF_a, (-1,), newLambda_a, (-1,)), gap), "getRho__")
    Gnu = (tprod(G_u, (-1, 1), nu, (-1,)) if nG > 0 else Tzeros((nU,)))
    if smallerNewtonMatrix:                                                 # :249-333
        if nF == 0:
            raise ValueError("smallerNewtonMatrix needs inequality constraints")
        LPG = tprod(rdivide(lam, F), (1,), F_u, (1, 2))
        WW11 = Lf_uu + tprod(F_u, (-1, 1), LPG, (-1, 2)) + addEye2HessianU * Teye((nU, nU))
        rows = [[WW11], []]
        if nG > 0:
            rows[0].append(transpose(G_u))
            rows[1] = [G_u, -(addEye2HessianEq * Teye((nG, nG)))]
        WW = blocks(rows)
        muF = rdivide(muOnes, F)
        factor_ww = ldl(WW, classname + "_WW", stages=[0] * nU + [1] * nG)
        tail = [-G] if nG > 0 else []
        if skipAffine:
            b_s = cat(1, -f_u - Gnu + tprod(F_u, (-1, 1), muF, (-1,)), *tail)
        else:
            b_a = cat(1, -f_u - Gnu, *tail)
            dx_a = code.declareAlias(mldivide(factor_ww, b_a), "dUNu_a__")
            dU_a = dx_a[0:nU]
            dLambda_a = -(LPG * dU_a) - lam
            affine(dU_a, dLambda_a)
            Mehrotra = rdivide(times(F_u * dU_a, dLambda_a), F)
            b_s = cat(1, -f_u - Gnu + tprod(F_u, (-1, 1), muF, (-1,)) - tprod(F_u, (-1, 1), Mehrotra, (-1,)), *tail)
        dx_s = code.declareAlias(mldivide(factor_ww, b_s), "dx_s__")
        dU_s = dx_s[0:nU]
        dLambda_s = muF - LPG * dU_s - lam
        if not skipAffine:
            dLambda_s = dLambda_s - Mehrotra
    else:                                                                   # :334-427
        WW11 = Lf_uu + addEye2HessianU * Teye((nU, nU))
        rows = [[WW11], [], []]
        if nG > 0:
            rows[0].append(transpose(G_u))
            rows[1] = [G_u, -(addEye2HessianEq * Teye((nG, nG)))]
        if nF > 0:
            rows[0].append(-transpose(F_u))
            if nG > 0:
                rows[1].append(Tzeros((nG, nF)))
            rows[2] = [-F_u] + ([Tzeros((nF, nG))] if nG > 0 else []) + [-diag(rdivide(F, lam))]
        WW = blocks(rows)
        factor_ww = ldl(WW, classname + "_WW", stages=[0] * nU + [1] * nG + [2] * nF)
        mid = [-G] if nG > 0 else []
        if skipAffine or nF == 0:
            b_s = cat(1, -Lf_u, *mid, *([F - rdivide(muOnes, lam)] if nF > 0 else []))
        else:
            b_a = cat(1, -Lf_u, *mid, F)
            dx_a = code.declareAlias(mldivide(factor_ww, b_a), "dUNu_a__")
            dU_a = dx_a[0:nU]
            dLambda_a = dx_a[nU + nG:nU + nG + nF]
            affine(dU_a, dLambda_a)
            b_s = cat(1, -Lf_u, *mid, F + rdivide(times(F_u * dU_a, dLambda_a), lam) - rdivide(muOnes, lam))
        dx_s = code.declareAlias(mldivide(factor_ww, b_s), "dx_s__")
        dU_s = dx_s[0:nU]
        dLambda_s = dx_s[nU + nG:nU + nG + nF] if nF > 0 else None
    out["Hess"] = WW
    out["dHess"], out["lHess"] = ldl_d(factor_ww), ldl_l(factor_ww)
    code.declareGet([tsum(compose(out["dHess"], "heaviside")), tsum(compose(-out["dHess"], "heaviside"))],
                    "getHessInertia__")
    code.declareGet([norminf(WW * dx_s - b_s)], "getDirectionError__")
    code.declareGet(dU_s * (WW11 * dU_s), "getCurvature__")
    dsts, srcs = [u], [u + alphaPrimal * dU_s]
    if nG > 0:
        dsts.append(nu)
        srcs.append(nu + alphaDualEq * dx_s[nU:nU + nG])
    if nF > 0:
        newU_s = u + alphaPrimal * dU_s
        code.declareGet([clp(F, F_u * dU_s), clp(lam, dLambda_s)], "getMaxAlphas_s__")
        code.declareGet(tmin(substitute(F, u, newU_s), 1), "getMinF_s__")
        dsts.append(lam)
        srcs.append(lam + alphaDualIneq * dLambda_s)                         # ADDITIVE update (:315,408)
    code.declareCopy(dsts, srcs, "updatePrimalDual__")
    out["b"], out["dx"] = b_s, dx_s
    # cost sensitivity with respect to some of the optimization variables (ipmPD_CS.m:478-503): with u = (u1, u2),
    # u1 the sensitivity variables, the other unknowns z = (u2, nu, lambda) solve Hess1*dz = B1*du1 around the
    # solution: Du1 = dz/du1, DfDu1 = Lf_u1, D2fDu1 = WW(u1,u1) + WW(u1,z)*Du1.  Outputs only ('Du1_', 'DfDu1_',
    # 'D2fDu1_' in outputExpressions); nothing of the Newton iteration changes.
    isSensitivity = np.zeros(nU, dtype=bool) if isSensitivity is None else np.asarray(isSensitivity, dtype=bool)
    if isSensitivity.shape != (nU,):
        raise ValueError("isSensitivity: one flag per packed optimization variable expected")
    if isSensitivity.any():
        if smallerNewtonMatrix:
            raise NotImplementedError("ipmPD_CS: computation of sensitivity not implemented for smallerNewtonMatrix "
                                      "(ipmPD_CS.m:479-481)")
        s1 = [int(k) for k in np.flatnonzero(isSensitivity)]
        rest = [int(k) for k in np.flatnonzero(~isSensitivity)] + list(range(nU, nU + nG + nF))
        out["DfDu1"] = Lf_u[s1]
        D2 = WW[s1, s1]
        if (~isSensitivity).any():
            out["Hess1"] = WW[rest, rest]
            out["B1"] = -WW[rest, s1]
            nr = int((~isSensitivity).sum())
            factor_Hess1 = ldl(out["Hess1"], classname + "_Hess1", stages=[0] * nr + [1] * nG + [2] * nF)
            out["Du1"] = mldivide(factor_Hess1, out["B1"])
            D2 = D2 + WW[s1, rest] * out["Du1"]
        else:
            out["Du1"] = Tzeros((0, len(s1)))
        out["D2fDu1"] = D2
    return out
