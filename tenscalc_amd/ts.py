"""Time-series helpers needed to state the MPC configs (scalar-``Ts`` branch only; SURVEY.md
Appendix B).  Mirrors /root/reference/lib/tsIntegral.m:58-61, tsDerivative.m:81-108,
tsDerivative2.m:64-81."""
from __future__ import annotations

import numpy as np

from . import tcalculus as tc


def tsIntegral(x, Ts):
    """trapezoidal integral over the last dimension: sum_k w_k x(...,k), w=Ts*[.5,1,...,1,.5]"""
    x = tc.toT(x)
    T = x.size[-1]
    w = np.ones(T)
    w[0] = w[-1] = .5
    dt = tc.toT(Ts) * tc.Tconstant(w)
    ix = tuple(range(1, x.ndim)) + (-1,)
    return tc.tprod(dt, (-1,), x, ix)


def tsDerivative(x, Ts):
    """3-point first derivative along the last dimension, same size as x (tsDerivative.m:81-108)"""
    x = tc.toT(x)
    T = x.size[-1]
    D = np.zeros((T, T))               # y(:,k) = sum_j x(:,j) D[j,k] / Ts
    D[0, 0], D[1, 0], D[2, 0] = -1.5, 2.0, -.5
    for k in range(1, T - 1):
        D[k - 1, k], D[k + 1, k] = -.5, .5
    D[T - 3, T - 1], D[T - 2, T - 1], D[T - 1, T - 1] = .5, -2.0, 1.5
    ix = tuple(range(1, x.ndim)) + (-1,)
    return tc.tprod(x, ix, tc.Tconstant(D), (-1, x.ndim)) / tc.toT(Ts)


def tsDerivative2(x, Ts):
    """second difference along the last dimension (tsDerivative2.m:64-81): columns 1,2 and T-1,T
    repeat the nearest interior stencil"""
    x = tc.toT(x)
    T = x.size[-1]
    D = np.zeros((T, T))
    for k in range(T):
        c = min(max(k, 1), T - 2)
        D[c - 1, k], D[c, k], D[c + 1, k] = 1.0, -2.0, 1.0
    ix = tuple(range(1, x.ndim)) + (-1,)
    Ts = tc.toT(Ts)
    return tc.tprod(x, ix, tc.Tconstant(D), (-1, x.ndim)) / (Ts * Ts)


def tsCross(x1, x2):
    """column-wise cross product of two 3xT series (lib/tsCross.m:41-45)"""
    x1, x2 = tc.toT(x1), tc.toT(x2)
    return (tc.times(x1[[1, 2, 0], :], x2[[2, 0, 1], :]) - tc.times(x1[[2, 0, 1], :], x2[[1, 2, 0], :]))
