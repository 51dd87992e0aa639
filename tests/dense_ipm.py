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


# ---- general nonlinear problems: derivatives by torch autograd, NOT by tenscalc_amd/tcalculus.py ------------------
def solve_nlp_dense(model, u0, mu0, maxIter, eps0=(1e-9, 1e-9), gradTol=1e-4, eqTol=1e-4, gapTol=1e-5, alphaMin=1e-7,
                    alphaMax=1.0, muAgg=.2, muCons=.95, epsTol=1e-6, trace=None):
    """The same method for  min f(u)  s.t.  G(u) = 0,  Fraw(u) >= 0  with f, G, Fraw given as torch functions of the
    packed variable (``model`` = dict(f=, G=, F=)): gradient, Jacobians and the Hessian of the Lagrangian come from
    torch.autograd in fp64, the KKT system is solved by dense LU.  Formulas: lib/ipmPD_CStimesLambda.m:86-281 (with the
    Hessian of the constraints in Lf_uu, which the QP version above does not need); decisions: lib/ipmPD_CSsolver.c:204-867
    including the regularisation retry on negative curvature (:455-531, useInertia=0).
    -> (status, iter, u, nu, lam); trace receives (norminf_grad, norminf_eq, gap, mu, alphaPrimal, curvature, epsU)"""
    import torch
    from torch.autograd.functional import hessian, jacobian
    t = lambda a: torch.as_tensor(np.asarray(a, dtype=float), dtype=torch.float64)        # noqa: E731
    f_, G_, F_ = model["f"], model["G"], model["F"]
    u = np.array(u0, dtype=float)
    n = len(u)
    nG, nF = len(G_(t(u))), len(F_(t(u)))
    s = np.abs(1.0 / F_(t(u)).numpy())                           # scaleIneq__, frozen once per solve (:86-89)
    st = t(s)
    F = lambda v: s * F_(t(v)).numpy()                           # noqa: E731
    mu = mu0
    muMin = gapTol / nF / 2
    lam = mu / F(u)                                              # initDualIneq__
    epsU, epsEq = eps0

    def derivs(v, lam_, nu_):
        vt = t(v)
        fu = jacobian(f_, vt).numpy()
        Gu = jacobian(G_, vt).numpy() if nG else np.zeros((0, n))
        Fu = s[:, None] * jacobian(F_, vt).numpy()
        lt, nt = t(lam_), t(nu_)
        L = lambda w: f_(w) - (lt * st * F_(w)).sum() + ((nt * G_(w)).sum() if nG else 0.0)   # noqa: E731
        H = hessian(L, vt).numpy()
        return fu, Gu, Fu, .5 * (H + H.T)
    fu, Gu, Fu, _ = derivs(u, lam, np.zeros(nG))
    if nG:                                                       # initDualEqX__ (:207-217)
        W0 = np.block([[np.eye(n), Gu.T], [Gu, -epsEq * np.eye(nG)]])
        nu = np.linalg.solve(W0, np.r_[Fu.T @ lam - fu, np.zeros(nG)])[n:]
    else:
        nu = np.zeros(0)
    updU = updEq = False
    newU, newEq = epsU, epsEq
    it = 0
    aP = aDI = aDE = 0.0
    curv = np.nan
    while True:
        it += 1
        fu, Gu, Fu, H = derivs(u, lam, nu)
        g = fu - Fu.T @ lam + (Gu.T @ nu if nG else 0)
        ng = np.abs(g).max()
        Gv = G_(t(u)).numpy() if nG else np.zeros(0)
        neq = np.abs(Gv).max() if nG else 0.0
        Fv = F(u)
        gap, ineq, dual = lam @ Fv, Fv.min(), lam.min()
        if trace is not None:
            trace.append((ng, neq, gap, mu, aP, curv, epsU))
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
        rhs = np.r_[-g, -Gv, lam * Fv - mu]

        def direction(eU, eEq):
            WW11 = H + eU * np.eye(n)
            WW = np.zeros((n + nG + nF, n + nG + nF))
            WW[:n, :n] = WW11
            if nG:
                WW[:n, n:n + nG] = Gu.T
                WW[n:n + nG, :n] = Gu
                WW[n:n + nG, n:n + nG] = -eEq * np.eye(nG)
            WW[:n, n + nG:] = -(Fu.T * lam)
            WW[n + nG:, :n] = -(Fu.T * lam).T
            WW[n + nG:, n + nG:] = -np.diag(Fv * lam)
            dx = np.linalg.solve(WW, rhs)
            return dx, dx[:n] @ (WW11 @ dx[:n]), np.abs(WW @ dx - rhs).max()
        dx, curv, derr = direction(epsU, epsEq)
        newU, newEq = epsU, epsEq
        if curv > 0:                                            # :425-447, applied at the next iteration
            if epsU > 1e-20 and derr < 1e-6:
                newU, updU = max(.75 * epsU, 1e-20), True
            if epsEq > 1e-20 and derr < 1e-6:
                newEq, updEq = max(.75 * epsEq, 1e-20), True
            if epsU < 1e2 and derr > 1e-6:
                newU, updU = min(10 * epsU, 1e2), True
            if epsEq < 1e2 and derr > 1e-6:
                newEq, updEq = min(10 * epsEq, 1e2), True
        elif curv < 0:                                          # :455-531: one retry with 10x the regularisation, now
            change = False
            if epsU < 1e2:
                epsU, change = min(10 * max(epsU, 1e-20), 1e2), True
            if epsEq < 1e2:
                epsEq, change = min(10 * max(epsEq, 1e-20), 1e2), True
            if change:
                dx, curv, derr = direction(epsU, epsEq)
            newU, newEq = epsU, epsEq
        dU, dNu, dL = dx[:n], dx[n:n + nG], dx[n + nG:]

        def clp(x, dd):
            m = dd < 0
            return (-x[m] / dd[m]).min() if m.any() else DBL_MAX
        aP = min(clp(Fv, Fu @ dU), clp(np.ones(nF), dL)) * .99
        aMax = min(aP, alphaMax)
        minF = lambda a: F(u + a * dU).min()                    # noqa: E731
        if aMax >= alphaMin:                                    # :692-756
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
        aDE = aDI = aP                                          # coupledAlphas
        u = u + aP * dU
        if nG:
            nu = nu + aDE * dNu
        lam = lam * (1 + aDI * dL)
        th_grad = ng < max(1e-6, gradTol)
        th_eq = (neq < max(1e-5, eqTol)) if nG else True
        if aP > aMax / 2 and th_grad and th_eq:                 # :782-847
            mu = max(mu * min(muAgg, mu ** .5), muMin)
        elif aP < .1:
            mu = min(mu * 1.1, mu0)
            lam = mu / F(u)
        elif aP > .99 and th_eq:
            mu = max(mu * muCons, muMin)
        if aP < alphaMin and aDI < alphaMin and aDE < alphaMin:
            mu = max(mu / muCons ** 2, muMin)


def quadcopter_model(d, T=100):
    """examples/mpcmhe/mpc_quadcopter.m:86-134 by hand, as torch functions of the packed variable
    [p(:) (3xT column-major), u(:) (3xT), positive2(T)]:
      v = tsDerivative(p,Ts) (lib/tsDerivative.m:81-108), a = tsDerivative2(p,Ts) (lib/tsDerivative2.m:64-81),
      equalities  a + b*v - g - u = 0,  p(:,1) = pinit,  v(:,1) = vinit,  max_thrust - |u_k| - positive2_k = 0;
      inequalities |u_k| - min_thrust >= 0,  positive2 >= 0,  -min_altitude - p_z >= 0;
      J = int |p - pdesired|^2 + lambda_v int |v - vdesired|^2 + lambda_thrust int |u|  (trapezoid, lib/tsIntegral.m:58-61)"""
    import torch
    t = lambda a: torch.as_tensor(np.asarray(a, dtype=float), dtype=torch.float64)        # noqa: E731
    Ts, b = float(d["Ts"][0]), float(d["b_drag"][0])
    pinit, vinit, pdes, vdes = (t(d[k]).reshape(3, 1) for k in ("pinit", "vinit", "pdesired", "vdesired"))
    gvec = t([0., 0., 9.8]).reshape(3, 1)
    w = torch.full((T,), Ts, dtype=torch.float64)
    w[0] = w[-1] = Ts / 2

    def split(z):
        p = z[:3 * T].reshape(T, 3).T                 # column-major 3xT
        u = z[3 * T:6 * T].reshape(T, 3).T
        return p, u, z[6 * T:]

    def vel(p):
        first = (-1.5 * p[:, 0:1] + 2 * p[:, 1:2] - .5 * p[:, 2:3]) / Ts
        mid = (.5 * p[:, 2:] - .5 * p[:, :-2]) / Ts
        last = (.5 * p[:, T - 3:T - 2] - 2 * p[:, T - 2:T - 1] + 1.5 * p[:, T - 1:T]) / Ts
        return torch.cat([first, mid, last], 1)

    def acc(p):
        mid = (p[:, :-2] - 2 * p[:, 1:-1] + p[:, 2:]) / Ts ** 2
        return torch.cat([mid[:, 0:1], mid, mid[:, -1:]], 1)

    def thrust(u):
        return torch.sqrt((u * u).sum(0))

    def f(z):
        p, u, _ = split(z)
        v = vel(p)
        return (w * ((p - pdes) ** 2).sum(0)).sum() + float(d["lambda_v"][0]) * (w * ((v - vdes) ** 2).sum(0)).sum() \
            + float(d["lambda_thrust"][0]) * (w * thrust(u)).sum()

    def G(z):
        p, u, pos2 = split(z)
        v = vel(p)
        dyn = acc(p) - (-(b * v) + gvec + u)
        return torch.cat([dyn.T.reshape(-1), (p[:, 0:1] - pinit).reshape(-1), (v[:, 0:1] - vinit).reshape(-1),
                          float(d["max_thrust"][0]) - thrust(u) - pos2])

    def F(z):
        p, u, pos2 = split(z)
        return torch.cat([thrust(u) - float(d["min_thrust"][0]), pos2, -float(d["min_altitude"][0]) - p[2, :]])
    u0 = np.r_[np.asarray(d["p"], float).reshape(-1, order="F"), np.asarray(d["u"], float).reshape(-1, order="F"),
               np.asarray(d["positive2"], float)]
    return dict(f=f, G=G, F=F), u0
