"""MPC for a DC motor: /root/reference/examples/mpcmhe/mpc_dcmotor.m:43-142 (BASELINE configs[3]):
horizon T=30, 2 states, 1 control; forward-Euler dynamics as equality constraints, box constraints
on future states and all controls; nU=89, nG=60, nF=174 -> 323x323 KKT.
Solve arguments mu0=1e-3, maxIter=100 (:152-153)."""
from __future__ import annotations

import numpy as np

from .. import tcalculus as tc
from ..cmex2optimizeCS import cmex2optimizeCS
from ..ts import tsIntegral

SOLVE_ARGS = dict(mu0=1e-3, maxIter=100, saveIter=-1, addEye2Hessian=(1e-9, 1e-9))
PARAMS = dict(Ts=.1, p=-2.0, k=1.0, min_x=[-.4, -.3], max_x=[.4, .3], min_u=-1.0, max_u=1.0,
              lambda_u=1 / 50)       # :163-171


def matrix(rows):
    """[a,b;c,d] from scalars / scalar expressions"""
    return tc.cat(1, *[tc.cat(2, *[tc.reshape(tc.toT(e), (1, 1)) for e in r]) for r in rows])


def mpc_dcmotor(T=30, classname="tmp_mpc_dcmotor", **options):
    nX, nU = 2, 1
    Ts = tc.Tvariable("Ts", ())
    x = tc.Tvariable("x", (nX, T))
    u = tc.Tvariable("u", (nU, T - 1))
    xinit = tc.Tvariable("xinit", (nX, 1))
    p = tc.Tvariable("p", ())
    k = tc.Tvariable("k", ())
    A = matrix([[0.0, 1.0], [0.0, p]])
    B = matrix([[0.0], [k]])
    dynamics = [x[:, 1:] == x[:, :-1] + Ts * (A * x[:, :-1] + B * u),
                x[:, 0:1] == xinit]
    min_x = tc.Tvariable("min_x", (nX, 1))
    max_x = tc.Tvariable("max_x", (nX, 1))
    min_u = tc.Tvariable("min_u", (nU, 1))
    max_u = tc.Tvariable("max_u", (nU, 1))
    constraints = [x[:, 1:] >= tc.repmat(min_x, (1, T - 1)), x[:, 1:] <= tc.repmat(max_x, (1, T - 1)),
                   u >= tc.repmat(min_u, (1, T - 1)), u <= tc.repmat(max_u, (1, T - 1))]
    ref = tc.Tvariable("ref", (1, T))
    lambda_u = tc.Tvariable("lambda_u", ())
    Jx2 = tsIntegral(tc.tsum(tc.sqr(x[0:1, :] - ref), 1), Ts)
    Ju2 = tsIntegral(tc.tsum(tc.sqr(u), 1), Ts)
    J = Jx2 + lambda_u * Ju2
    uWarm = tc.cat(2, u[:, 1:], tc.Tzeros((nU, 1)))
    xWarm = x[:, list(range(1, T)) + [T - 1]]
    xWarm = tc.max2(xWarm, tc.repmat(min_x + .05 * (max_x - min_x), (1, T)))
    xWarm = tc.min2(xWarm, tc.repmat(max_x - .05 * (max_x - min_x), (1, T)))
    uWarm = tc.max2(uWarm, tc.repmat(min_u + .05 * (max_u - min_u), (1, T - 1)))
    uWarm = tc.min2(uWarm, tc.repmat(max_u - .05 * (max_u - min_u), (1, T - 1)))
    outs = dict(J=J, Jx2=Jx2, Ju2=Ju2, t=tc.Tconstant(np.arange(T, dtype=float)) * Ts, u=u, x=x, ref=ref,
                xWarm=tc.full(xWarm), uWarm=tc.full(uWarm))
    # Elimination order of the two KKT factorisations (the reference calls symamd, which is not available: DESIGN.md 7).
    # Round 5: multiple elimination with degree tolerance 1 -- 177 -> 51 dependent levels in the factor, 440 -> 246 per Newton
    # iteration over factor + solves, 1.06x the fill; every back-end that walks levels gains (LDS machine 34.0 -> 21.5 ms per
    # 32768, tile machine 23.0 -> 20.9: profiles/r05_dcmotor_orders.log).  Numerically the two orders are the same: direction
    # error 4.5e-15 against 3.8e-15 at worst over 60 seeds x 4 stopping points, and either order leaves the tape-independent
    # dense IPM through a line-search branch (DESIGN.md 4 "knife edge") on 9 / 10 of those 60 seeds -- different ones
    # (tests/test_tape_vs_dense_ipm.py pins the seeds that agree and asserts the rates).  Tolerance >= 2 does lose digits on the
    # equality duals (DESIGN.md 7 "numerical finding").
    options.setdefault("ldlOrder", ("mmd", 1))
    return cmex2optimizeCS(classname=classname, objective=J, optimizationVariables=[u, x],
                           constraints=dynamics + constraints, outputExpressions=list(outs.values()),
                           outputNames=list(outs.keys()),
                           parameters=[Ts, p, k, xinit, ref, min_x, max_x, min_u, max_u, lambda_u],
                           addEye2Hessian=True, adjustAddEye2Hessian=True, scaleInequalities=True, **options)


def dcmotor_data(seed, T=30):
    """one instance of the parameter sweep of SURVEY.md 8d config 4: xinit uniform in
    [-.35,.35]x[-.25,.25], reference phase t0~U(0,4pi) in ref(t)=-.35*sign(sin(.5t)) (:161,195),
    cold start xWarm=xinit+.01*rand, uWarm=.01*rand (:186-187)."""
    rs = np.random.RandomState(seed)
    xinit = np.array([-.35 + .7 * rs.random_sample(), -.25 + .5 * rs.random_sample()])
    t0 = 4 * np.pi * rs.random_sample()
    t = t0 + np.arange(T) * PARAMS["Ts"]
    ref = -.35 * np.sign(np.sin(.5 * t))
    xWarm = xinit[:, None] + .01 * rs.random_sample(2 * T).reshape((2, T), order="F")
    uWarm = .01 * rs.random_sample(T - 1).reshape((1, T - 1), order="F")
    d = {k: np.atleast_1d(np.asarray(v, dtype=float)) for k, v in PARAMS.items()}
    d.update(xinit=xinit, ref=ref, x=xWarm, u=uWarm)
    return d
