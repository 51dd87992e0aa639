"""Quadcopter MPC: /root/reference/examples/mpcmhe/mpc_quadcopter.m:64-182 (BASELINE configs[4]) as
written in the file: horizon T=100, 3-D position model, variables p[3,T], u[3,T], positive2[T]
(nU=700), nG=406 equality rows (dynamics via tsDerivative/tsDerivative2 + initial conditions +
thrust slack), nF=300 -> 1406x1406 KKT.  mu0=1e-3, maxIter=1000 (:189-190)."""
from __future__ import annotations

import numpy as np

from .. import tcalculus as tc
from ..cmex2optimizeCS import cmex2optimizeCS
from ..ts import tsCross, tsDerivative, tsDerivative2, tsIntegral

SOLVE_ARGS = dict(mu0=1e-3, maxIter=1000, saveIter=-1, addEye2Hessian=(1e-9, 1e-9))
PARAMS = dict(Ts=.02, b_drag=.1, min_altitude=-.1, min_thrust=5.0, max_thrust=20.0, lambda_v=.05,
              lambda_thrust=.05)     # :200-213


def mpc_quadcopter(T=100, classname="tmp_quadcopter", **options):
    Ts = tc.Tvariable("Ts", ())
    p = tc.Tvariable("p", (3, T))
    u = tc.Tvariable("u", (3, T))
    pinit = tc.Tvariable("pinit", (3, 1))
    vinit = tc.Tvariable("vinit", (3, 1))
    b_drag = tc.Tvariable("b_drag", ())
    g = tc.Tconstant(np.array([[0.], [0.], [9.8]]), (3, 1))
    min_thrust = tc.Tvariable("min_thrust", ())
    max_thrust = tc.Tvariable("max_thrust", ())
    min_altitude = tc.Tvariable("min_altitude", ())
    v = tsDerivative(p, Ts)
    a = tsDerivative2(p, Ts)
    dynamics = [a == -(b_drag * v) + tc.repmat(g, (1, T)) + u,
                p[:, 0:1] == pinit,
                v[:, 0:1] == vinit]
    m_thrust = tc.sqrt(tc.tsum(tc.sqr(u), 1))
    R_ib3 = u / tc.repmat(tc.reshape(m_thrust, (1, T)), (3, 1))
    R_ib2 = tsCross(R_ib3, tc.cat(1, tc.Tzeros((2, T)), tc.Tones((1, T))))
    R_ib1 = tsCross(R_ib2, R_ib3)
    positive2 = tc.Tvariable("positive2", (T,))
    constraints = [m_thrust >= min_thrust,
                   max_thrust - m_thrust == positive2, positive2 >= 0,
                   tc.reshape(p[2:3, :], (T,)) <= -min_altitude]
    pdesired = tc.Tvariable("pdesired", (3, 1))
    vdesired = tc.Tvariable("vdesired", (3, 1))
    lambda_v = tc.Tvariable("lambda_v", ())
    lambda_thrust = tc.Tvariable("lambda_thrust", ())
    Jp2 = tsIntegral(tc.tsum(tc.sqr(p - tc.repmat(pdesired, (1, T))), 1), Ts)
    Jv2 = tsIntegral(tc.tsum(tc.sqr(v - tc.repmat(vdesired, (1, T))), 1), Ts)
    Jthrust = tsIntegral(m_thrust, Ts)
    J = Jp2 + lambda_v * Jv2 + lambda_thrust * Jthrust
    outs = dict(J=J, Jp2=Jp2, Jv2=Jv2, Jthrust=Jthrust, t=tc.Tconstant(np.arange(T, dtype=float)) * Ts, u=u,
                m_thrust=m_thrust, p=p, v=v, a=a, R_ib1=R_ib1, R_ib2=R_ib2, R_ib3=R_ib3, positive2=positive2)
    # Elimination order of the KKT factorisations (the reference calls symamd, which is not available: DESIGN.md 7).
    # Round 5: multiple elimination with degree tolerance 0 (independent sets of minimum-degree nodes per round): the
    # chain through factor + solves is 2.3x shorter at 1.03x the fill, and the pivots are those of the greedy order in
    # quality (smallest pivot 5.6e-6, solve error 1.2e-6 on the KKT matrix at the initial point).  Round 3 kept
    # the greedy order because a batch of 512 ends with its slowest instance and this order moves that instance from 367
    # to 563 iterations; at configs[4]'s stated batch of 4096 on one GPU the CUs are handed the next instance when one
    # finishes, so what counts is the time per Newton iteration: 614 -> 522 ms per 4096 (profiles/r05_quadcopter_orders.log).
    # The tape-independent dense NLP pin holds unchanged (tests/test_tape_vs_dense_ipm.py: same 26 iterations, 1e-10).
    options.setdefault("ldlOrder", ("mmd", 0))
    return cmex2optimizeCS(classname=classname, objective=J, optimizationVariables=[p, u, positive2],
                           constraints=dynamics + constraints, outputExpressions=list(outs.values()),
                           outputNames=list(outs.keys()),
                           parameters=[Ts, pinit, vinit, pdesired, vdesired, b_drag, min_thrust, max_thrust,
                                       min_altitude, lambda_v, lambda_thrust],
                           addEye2Hessian=True, adjustAddEye2Hessian=True, scaleInequalities=True, **options)


def _deriv(p, Ts):
    T = p.shape[1]
    v = np.empty_like(p)
    v[:, 0] = (-1.5 * p[:, 0] + 2 * p[:, 1] - .5 * p[:, 2]) / Ts
    v[:, 1:-1] = (.5 * p[:, 2:] - .5 * p[:, :-2]) / Ts
    v[:, -1] = (.5 * p[:, T - 3] - 2 * p[:, T - 2] + 1.5 * p[:, T - 1]) / Ts
    return v


def _deriv2(p, Ts):
    T = p.shape[1]
    a = np.empty_like(p)
    for k in range(T):
        c = min(max(k, 1), T - 2)
        a[:, k] = (p[:, c - 1] - 2 * p[:, c] + p[:, c + 1]) / Ts ** 2
    return a


def quadcopter_data(seed, T=100):
    """one instance of the sweep of SURVEY.md 8d config 5: pdesired on a sphere of radius <=5.6,
    vinit~U(-1,1)^3; spline initial guess of mpc_quadcopter.m:234-247."""
    from scipy.interpolate import CubicSpline
    rs = np.random.RandomState(seed)
    d = {k: np.array([float(v)]) for k, v in PARAMS.items()}
    Ts, b_drag = PARAMS["Ts"], PARAMS["b_drag"]
    pinit = np.zeros(3)
    vinit = -1 + 2 * rs.random_sample(3)
    vinit[2] = min(vinit[2], 0.0)                    # do not start by diving through the altitude floor
    dirn = rs.standard_normal(3)
    dirn[2] = -abs(dirn[2])                          # z is "down": keep the target above the floor
    pdesired = 5.6 * rs.random_sample() ** (1 / 3) * dirn / np.linalg.norm(dirn)
    pfinal = pinit + (pdesired - pinit) / 10
    t = np.arange(1, T + 1) * Ts
    pts = np.stack([pinit, pinit + Ts * vinit, pinit + 2 * Ts * vinit, pfinal], axis=1)
    cs = CubicSpline([t[0], t[1], t[2], t[-1]], pts, axis=1, bc_type="not-a-knot")
    p = cs(t)
    v, a = _deriv(p, Ts), _deriv2(p, Ts)
    u = a + b_drag * v - np.array([[0.], [0.], [9.8]])
    d.update(pinit=pinit, vinit=vinit, pdesired=pdesired, vdesired=np.zeros(3), p=p, u=u, positive2=np.ones(T))
    return d
