"""Tape-independent oracle for the two convex BASELINE problems with linear constraints (mpc_dcmotor, l1l2):
a dense numpy primal-dual interior-point method on

        min 1/2 u'Pu + q'u + r    s.t.  Eu + e = 0,  Cu + d >= 0

that follows the formulas of /root/reference/lib/ipmPD_CStimesLambda.m:86-281 (inequality / cost scaling, Lagrangian,
KKT matrix WW, right-hand side b, multiplicative lambda update, least-squares initialisation of nu) and the
decisions of lib/ipmPD_CSsolver.c:204-867 (exit tests, addEye2Hessian adaptation, line search, mu schedule) -- the
generalisation of tests/sls_closed_form.py to equality constraints and scaleCost.  The matrices P, q, r, E, e, C, d
are written down by hand from the problem statements (examples/mpcmhe/mpc_dcmotor.m:52-142,
examples/l1l2estimationCS.m:137-183), NOT derived through tenscalc_amd/tcalculus.py / csparse.py: agreement of the
scalar tape with this code pins TAPE CONSTRUCTION (differentiation, scalarisation, KKT assembly, symbolic LDL and
its elimination order, triangular solves) for those problems.  Test infrastructure only."""
import numpy as np

DBL_MAX = np.finfo(float).max


def solve_qp_dense(P, q, r, E, e, C, d, u0, mu0, maxIter, eps0=(1e-9, 1e-9), scaleCost=0.0, gradTol=1e-4, eqTol=1e-4,
                   gapTol=1e-5, alphaMin=1e-7, alphaMax=1.0, muAgg=.2, muCons=.95, epsTol=1e-6, trace=None):
    """-> (status, iter, u, nu, lam).  trace: list that receives (norminf_grad, norminf_eq, gap, mu, alpha) per pass"""
    n, nG, nF = len(q), E.shape[0], C.shape[0]
    u = np.array(u0, dtype=float)
    f = lambda u: .5 * u @ (P @ u) + q @ u + r          # noqa: E731
    sc = 1.0
    if scaleCost:
        sc = abs(scaleCost / f(u))                      # ipmPD_CStimesLambda.m:96-103
        gapTol = gapTol * sc
    Pm, qv = sc * P, sc * q
    s = np.abs(1.0 / (C @ u + d))                       # :86-89, frozen by scaleIneq__ once per solve
    Fu = s[:, None] * C
    F = lambda u: s * (C @ u + d)                       # noqa: E731
    mu = mu0
    muMin = gapTol / nF / 2
    lam = mu / F(u)                                     # initDualIneq__
    epsU, epsEq = eps0
    fu = Pm @ u + qv
    if nG:                                              # initDualEqX__ (:207-217)
        W0 = np.block([[np.eye(n), E.T], [E, -epsEq * np.eye(nG)]])
        nu = np.linalg.solve(W0, np.r_[Fu.T @ lam - fu, np.zeros(nG)])[n:]
    else:
        nu = np.zeros(0)
    updU = updEq = False
    newU, newEq = epsU, epsEq
    it = 0
    aP = aDI = aDE = 0.0
    while True:
        it += 1
        g = Pm @ u + qv - Fu.T @ lam + (E.T @ nu if nG else 0)
        ng = np.abs(g).max()
        neq = np.abs(E @ u + e).max() if nG else 0.0
        Fv = F(u)
        gap, ineq, dual = lam @ Fv, Fv.min(), lam.min()
        if trace is not None:
            trace.append((ng, neq, gap, mu, aP))
        if it > maxIter:
            return 8, it, u, nu, lam
        if np.isnan(ng):
            return 4, it, u, nu, lam
        if ineq <= 0:
            return 1, it, u, nu, lam
        if dual <= 0:
            return 2, it, u, nu, lam
        if ng <= gradTol and gap <= gapTol and neq <= eqTol and epsU <= epsTol:
            return 0, it, u, nu, lam
        if updU:
            epsU, updU = newU, False
        if updEq:
            epsEq, updEq = newEq, False
        WW11 = Pm + epsU * np.eye(n)
        WW = np.zeros((n + nG + nF, n + nG + nF))
        WW[:n, :n] = WW11
        if nG:
            WW[:n, n:n + nG] = E.T
            WW[n:n + nG, :n] = E
            WW[n:n + nG, n:n + nG] = -epsEq * np.eye(nG)
        WW[:n, n + nG:] = -(Fu.T * lam)
        WW[n + nG:, :n] = -(Fu.T * lam).T
        WW[n + nG:, n + nG:] = -np.diag(Fv * lam)
        rhs = np.r_[-g, -(E @ u + e) if nG else np.zeros(0), lam * Fv - mu]
        dx = np.linalg.solve(WW, rhs)
        dU, dNu, dL = dx[:n], dx[n:n + nG], dx[n + nG:]
        curv = dU @ (WW11 @ dU)
        derr = np.abs(WW @ dx - rhs).max()
        if not curv > 0:
            raise ArithmeticError("negative curvature on a convex problem")
        newU, newEq = epsU, epsEq                       # ipmPD_CSsolver.c:425-447: applied at the next iteration
        if epsU > 1e-20 and derr < 1e-6:
            newU, updU = max(.75 * epsU, 1e-20), True
        if epsEq > 1e-20 and derr < 1e-6:
            newEq, updEq = max(.75 * epsEq, 1e-20), True
        if epsU < 1e2 and derr > 1e-6:
            newU, updU = min(10 * epsU, 1e2), True
        if epsEq < 1e2 and derr > 1e-6:
            newEq, updEq = min(10 * epsEq, 1e2), True

        def clp(x, dd):
            m = dd < 0
            return (-x[m] / dd[m]).min() if m.any() else DBL_MAX
        aP = min(clp(Fv, Fu @ dU), clp(np.ones(nF), dL)) * .99
        aMax = min(aP, alphaMax)
        minF = lambda a: F(u + a * dU).min()            # noqa: E731
        if aMax >= alphaMin:                            # :692-756
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
        aDE = aDI = aP                                  # coupledAlphas
        u = u + aP * dU
        if nG:
            nu = nu + aDE * dNu
        lam = lam * (1 + aDI * dL)
        th_grad = ng < max(1e-6, gradTol)
        th_eq = (neq < max(1e-5, eqTol)) if nG else True
        if aP > aMax / 2 and th_grad and th_eq:         # :782-847
            mu = max(mu * min(muAgg, mu ** .5), muMin)
        elif aP < .1:
            mu = min(mu * 1.1, mu0)
            lam = mu / F(u)
        elif aP > .99 and th_eq:
            mu = max(mu * muCons, muMin)
        if aP < alphaMin and aDI < alphaMin and aDE < alphaMin:
            mu = max(mu / muCons ** 2, muMin)


# ---- the two problems, by hand ------------------------------------------------------------------------------------
def dcmotor_qp(d, T=30):
    """examples/mpcmhe/mpc_dcmotor.m:52-104 as a QP in the packed variable [u(1,1:T-1), x(:) column-major]:
    J = Ts*sum_k w_k (x1_k - ref_k)^2 + lambda_u*Ts*sum_k w'_k u_k^2 (trapezoid weights of tsIntegral),
    x(:,k+1) = x(:,k) + Ts*(A x(:,k) + B u_k), x(:,1) = xinit; boxes on x(:,2:T) and u."""
    Ts, p, k, lu = float(d["Ts"][0]), float(d["p"][0]), float(d["k"][0]), float(d["lambda_u"][0])
    nu_, n = T - 1, (T - 1) + 2 * T
    ix = lambda i, kk: nu_ + 2 * kk + i                   # noqa: E731  x(i,kk), 0-based, column-major
    wx = np.ones(T); wx[0] = wx[-1] = .5
    wu = np.ones(T - 1); wu[0] = wu[-1] = .5
    P = np.zeros((n, n)); q = np.zeros(n); r = 0.0
    for kk in range(T):
        P[ix(0, kk), ix(0, kk)] += 2 * Ts * wx[kk]
        q[ix(0, kk)] += -2 * Ts * wx[kk] * d["ref"][kk]
        r += Ts * wx[kk] * d["ref"][kk] ** 2
    for kk in range(T - 1):
        P[kk, kk] += 2 * lu * Ts * wu[kk]
    A = np.array([[0., 1.], [0., p]]); Bm = np.array([0., k])
    E = np.zeros((2 * (T - 1) + 2, n)); e = np.zeros(2 * (T - 1) + 2)
    row = 0
    for kk in range(T - 1):                               # x(:,kk+1) - (x(:,kk) + Ts*(A x(:,kk) + B u_kk)) = 0
        for i in range(2):
            E[row, ix(i, kk + 1)] += 1
            E[row, ix(i, kk)] -= 1
            for j in range(2):
                E[row, ix(j, kk)] -= Ts * A[i, j]
            E[row, kk] -= Ts * Bm[i]
            row += 1
    for i in range(2):
        E[row, ix(i, 0)] = 1; e[row] = -d["xinit"][i]; row += 1
    C, dd = [], []
    for sgn, lim in ((1, d["min_x"]), (-1, d["max_x"])):   # x(:,2:T) >= min_x ; max_x - x(:,2:T) >= 0
        for kk in range(1, T):
            for i in range(2):
                c = np.zeros(n); c[ix(i, kk)] = sgn
                C.append(c); dd.append(-sgn * lim[i])
    for sgn, lim in ((1, d["min_u"]), (-1, d["max_u"])):
        for kk in range(T - 1):
            c = np.zeros(n); c[kk] = sgn
            C.append(c); dd.append(-sgn * lim[0])
    u0 = np.r_[np.asarray(d["u"]).reshape(-1, order="F"), np.asarray(d["x"]).reshape(-1, order="F")]
    return P, q, r, E, e, np.array(C), np.array(dd), u0


def l1l2_qp(d, N=200):
    """examples/l1l2estimationCS.m:137-183: variables [position(N), noise1(N), acceleration1(N-2), noise1abs(N),
    acceleration1abs(N-2)]; J = |m - pos - n1|^2 + w2a |acc(pos) - a1|^2 + w1n sum(n1abs) + w1a sum(a1abs) with
    velocity = (pos(2:N)-pos(1:N-1)).*dt1, acc = (vel(2:end)-vel(1:end-1)).*dt1(1:end-1);
    constraints n1abs - n1 >= 0, n1 + n1abs >= 0, a1abs - a1 >= 0, a1 + a1abs >= 0."""
    m, dt1 = np.asarray(d["measurement"], float), np.asarray(d["dt1"], float)
    w1a, w2a, w1n = float(d["weight1acceleration"][0]), float(d["weight2acceleration"][0]), float(d["weight1noise"][0])
    n = N + N + (N - 2) + N + (N - 2)
    oP, oN, oA, oNa, oAa = 0, N, 2 * N, 3 * N - 2, 4 * N - 2
    V = np.zeros((N - 1, N))
    for k in range(N - 1):
        V[k, k + 1], V[k, k] = dt1[k], -dt1[k]
    D = np.zeros((N - 2, N))
    for k in range(N - 2):
        D[k] = (V[k + 1] - V[k]) * dt1[k]
    R1 = np.zeros((N, n)); R1[:, oP:oP + N] = -np.eye(N); R1[:, oN:oN + N] = -np.eye(N)       # noise2 = m + R1 u
    R2 = np.zeros((N - 2, n)); R2[:, oP:oP + N] = D; R2[:, oA:oA + N - 2] = -np.eye(N - 2)    # acceleration2 = R2 u
    P = 2 * (R1.T @ R1 + w2a * R2.T @ R2)
    q = 2 * R1.T @ m
    q[oNa:oNa + N] += w1n
    q[oAa:oAa + N - 2] += w1a
    r = m @ m
    C = np.zeros((2 * N + 2 * (N - 2), n)); row = 0
    for k in range(N):
        C[row, oNa + k], C[row, oN + k] = 1, -1; row += 1
    for k in range(N):
        C[row, oN + k], C[row, oNa + k] = 1, 1; row += 1
    for k in range(N - 2):
        C[row, oAa + k], C[row, oA + k] = 1, -1; row += 1
    for k in range(N - 2):
        C[row, oA + k], C[row, oAa + k] = 1, 1; row += 1
    u0 = np.r_[d["position"], d["noise1"], d["acceleration1"], d["noise1abs"], d["acceleration1abs"]]
    return P, q, r, np.zeros((0, n)), np.zeros(0), C, np.zeros(C.shape[0]), np.asarray(u0, float)
