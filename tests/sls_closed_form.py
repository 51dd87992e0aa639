"""Third, tape-independent oracle for sls: the dense closed form of SURVEY.md Appendix A
(derived from examples/sls.m:136-152 + ipmPD_CStimesLambda.m) written with numpy, running the
same Newton-loop decisions (ipmPD_CSsolver.c:282-867).  Used to pin TAPE CONSTRUCTION (gradient,
Hessian, KKT assembly, LDL) of the scalar tape: both must produce the same iterates up to
rounding."""
import numpy as np


def solve_sls_dense(A, b, x0, mu0=1.0, maxIter=30, eps0=1e-9, box=.05, gradTol=1e-4, gap_tol=1e-5,
                    alphaMin=1e-7, alphaMax=1.0, muAgg=.2, muCons=.95, epsTol=1e-6):
    N, n = A.shape
    nF = 2 * n
    cN = 1.0 / N
    H0 = 2 * cN * A.T @ A
    u = x0.copy()

    def Fraw(u):
        return np.r_[u, box - u]
    s = np.abs(1 / Fraw(u))
    Fu = np.vstack([np.diag(s[:n]), -np.diag(s[n:])])

    def F(u):
        return s * Fraw(u)
    mu = mu0
    muMin = gap_tol / nF / 2
    lam = mu / F(u)
    epsU = eps0
    upd = False
    it = 0
    trace = []
    while True:
        it += 1
        if it > maxIter:
            return 8, it, u, trace
        g = 2 * cN * A.T @ (A @ u - b) - Fu.T @ lam
        ng = np.abs(g).max()
        Fv = F(u)
        gap, ineq, dual = lam @ Fv, Fv.min(), lam.min()
        trace.append((ng, gap, mu))
        if ineq <= 0:
            return 1, it, u, trace
        if dual <= 0:
            return 2, it, u, trace
        if ng <= gradTol and gap <= gap_tol and epsU <= epsTol:
            return 0, it, u, trace
        if upd:
            upd = False
        WW11 = H0 + epsU * np.eye(n)
        WW = np.block([[WW11, -Fu.T * lam], [-(Fu.T * lam).T, -np.diag(Fv * lam)]])
        rhs = np.r_[-g, lam * Fv - mu]
        dx = np.linalg.solve(WW, rhs)
        dU, dL = dx[:n], dx[n:]
        curv = dU @ (WW11 @ dU)
        derr = np.abs(WW @ dx - rhs).max()
        assert curv > 0
        if epsU > 1e-20 and derr < 1e-6:
            epsU_next = max(.75 * epsU, 1e-20)
        elif epsU < 1e2 and derr > 1e-6:
            epsU_next = min(10 * epsU, 1e2)
        else:
            epsU_next = epsU

        def clp(x, d):
            m = d < 0
            return (-x[m] / d[m]).min() if m.any() else np.finfo(float).max
        aP = min(clp(Fv, Fu @ dU), clp(np.ones(nF), dL)) * .99
        aMax = min(aP, alphaMax)

        def minF(a):
            return F(u + a * dU).min()
        if aMax >= alphaMin:
            aP = aMax / .99
            i0 = minF(aP)
            i1 = None
            if i0 > 0:
                aP *= .99
                i1 = minF(aP)
            if i0 <= 0 or i1 <= i0 / 10:
                aP = alphaMin / .99
                if minF(aP) > 0:
                    aP = aMax * .95
                    while aP >= alphaMin:
                        i0 = minF(aP)
                        if i0 > 0:
                            aP *= .99
                            if minF(aP) > i0 / 10:
                                break
                        aP /= 2
                    if aP < alphaMin:
                        aP = 0
                else:
                    aP = 0
        else:
            aP = 0
        u = u + aP * dU
        lam = lam * (1 + aP * dL)
        th_grad = ng < max(1e-6, gradTol)
        if aP > aMax / 2 and th_grad:
            mu *= min(muAgg, mu ** .5)
            mu = max(mu, muMin)
        elif aP < .1:
            mu = min(mu * 1.1, mu0)
            lam = mu / F(u)
        elif aP > .99:
            mu = max(mu * muCons, muMin)
        if aP < alphaMin:
            mu = max(mu / muCons ** 2, muMin)
        epsU = epsU_next
