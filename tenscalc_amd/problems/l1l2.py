"""Robust l1-l2 trajectory estimation: second problem of
/root/reference/examples/l1l2estimationCS.m:137-218 (BASELINE configs[2]): N=200 samples;
variables position[N], noise1[N], acceleration1[N-2], noise1abs[N], acceleration1abs[N-2]
(nU=996), four inequality blocks (nF=796) -> 1792x1792 KKT; scaleCost=1, desiredDualityGap=1e-3;
mu0=1, maxIter=100 (:214-218)."""
from __future__ import annotations

import numpy as np

from .. import tcalculus as tc
from ..cmex2optimizeCS import cmex2optimizeCS

SOLVE_ARGS = dict(mu0=1.0, maxIter=100, saveIter=-1, addEye2Hessian=(1e-9, 1e-9))


def l1l2(N=200, classname="tmpC_minl1l2", **options):
    measurement = tc.Tvariable("measurement", (N,))
    dt1 = tc.Tvariable("dt1", (N - 1,))
    weight2acceleration = tc.Tvariable("weight2acceleration", ())
    position = tc.Tvariable("position", (N,))
    weight1acceleration = tc.Tvariable("weight1acceleration", ())
    weight1noise = tc.Tvariable("weight1noise", ())
    noise1 = tc.Tvariable("noise1", (N,))
    acceleration1 = tc.Tvariable("acceleration1", (N - 2,))
    noise1abs = tc.Tvariable("noise1abs", (N,))
    acceleration1abs = tc.Tvariable("acceleration1abs", (N - 2,))
    velocity = tc.times(position[1:] - position[:-1], dt1)
    acceleration = tc.times(velocity[1:] - velocity[:-1], dt1[:-1])
    noise2 = measurement - position - noise1
    acceleration2 = acceleration - acceleration1
    J = (tc.norm2(noise2) + weight2acceleration * tc.norm2(acceleration2)
         + weight1noise * tc.tsum(noise1abs, 1) + weight1acceleration * tc.tsum(acceleration1abs, 1))
    # elimination order: the shortest dependency chain 64 parallel lanes can pay for (csparse.depth_aware_order) --
    # no equality constraints here, so every order is as accurate as the greedy one; the banded 1792x1792 KKT matrix
    # is then reduced odd-even instead of front to back (critical path 1878 -> 111 instruction levels)
    opt = dict(scaleCost=1.0, desiredDualityGap=1e-3, ldlOrder=("depth", 64))
    opt.update(options)
    return cmex2optimizeCS(
        classname=classname, objective=J,
        optimizationVariables=[position, noise1, acceleration1, noise1abs, acceleration1abs],
        constraints=[noise1 <= noise1abs, noise1 >= -noise1abs,
                     acceleration1 <= acceleration1abs, acceleration1 >= -acceleration1abs],
        outputExpressions=[J, position, noise1, acceleration1, noise1abs, acceleration1abs],
        outputNames=["J", "position", "noise1", "acceleration1", "noise1abs", "acceleration1abs"],
        parameters=[measurement, dt1, weight1acceleration, weight2acceleration, weight1noise], **opt)


def l1l2_data(seed, N=200):
    """data generator of l1l2estimationCS.m:34-48 (seed 1 there) and initial values :193-197"""
    rs = np.random.RandomState(seed)
    t = np.cumsum(np.ceil(1 * rs.random_sample(N)))
    truePosition = 5 * np.sin(np.abs(t - 100) / 5)
    meas = np.round(truePosition + 1.0 * rs.standard_normal(N))
    k = np.flatnonzero(rs.random_sample(N) < .1)
    meas[k] = np.round(10 * rs.standard_normal(len(k)))
    dt1 = 1. / (t[1:] - t[:-1])
    return dict(measurement=meas, dt1=dt1, weight1acceleration=np.array([1.]),
                weight2acceleration=np.array([10.]), weight1noise=np.array([1.]),
                position=rs.random_sample(N), noise1=rs.random_sample(N),
                acceleration1=rs.random_sample(N - 2), noise1abs=10 * np.ones(N),
                acceleration1abs=10 * np.ones(N - 2), truePosition=truePosition)
