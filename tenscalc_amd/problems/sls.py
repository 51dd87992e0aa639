"""Box-constrained least squares: /root/reference/examples/sls.m:136-166 with the README sizes
N=100, n=8 (README.md:166).  ``J=norm2(A*x-b)/N``, ``0<=x<=.05``, parameters ``{A,b}``,
outputs ``{J,x}``; ``mu0=1, maxIter=30`` (sls.m:160-163)."""
from __future__ import annotations

import numpy as np

from .. import tcalculus as tc
from ..cmex2optimizeCS import cmex2optimizeCS

SOLVE_ARGS = dict(mu0=1.0, maxIter=30, saveIter=-1, addEye2Hessian=(1e-9, 1e-9))


def sls(N=100, n=8, classname="tmp_minslsc2", **options):
    A = tc.Tvariable("A", (N, n))
    b = tc.Tvariable("b", (N,))
    x = tc.Tvariable("x", (n,))
    y = A * x - b
    J = tc.norm2(y) / N
    return cmex2optimizeCS(classname=classname, objective=J, optimizationVariables=[x],
                           constraints=[x >= 0, x <= .05], outputExpressions=[J, x],
                           outputNames=["J", "x"], parameters=[A, b], **options)


def sls_data(seed, N=100, n=8):
    """``thisA=rand(N,n); thisb=rand(N,1); x0=.02*rand(n,1)`` drawn in that order from
    MT19937 (sls.m:36-41); A is returned column-major-flattened as MATLAB stores it."""
    rs = np.random.RandomState(seed)
    A = rs.random_sample(N * n).reshape((N, n), order="F")
    b = rs.random_sample(N)
    x0 = .02 * rs.random_sample(n)
    return A, b, x0


def sls_batch(B, seed0=0, N=100, n=8):
    """instance k uses seed seed0+k; arrays are [B, ...] (batch slowest on the host)."""
    A = np.empty((B, N, n))
    b = np.empty((B, N))
    x0 = np.empty((B, n))
    for k in range(B):
        A[k], b[k], x0[k] = sls_data(seed0 + k, N, n)
    return A, b, x0
