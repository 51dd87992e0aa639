"""The three two-player games of /root/reference/examples/TCgames.m, stated for ``cmex2equilibriumLatentCS``:

* ``matrixsaddle`` (:30-76): mixed-policy saddle point of a matrix game, ``J1 = u'*A1*d``, player 1 minimises over the
  simplex ``{sum(u)==1, u>=0}``, player 2 minimises ``-J1`` over its simplex (N1=50, N2=20 there);
* ``qsaddle`` (:98-143): ``Ju = norm2(u+d-2) - 10*norm2(d)`` with box constraints (N=200 there);
* ``qsaddlestate`` (:146-200): the same through a latent variable ``x == u+d``.

``mu0=1, maxIter=30`` (:70-72)."""
from __future__ import annotations

import numpy as np

from .. import tcalculus as tc
from ..equilibrium import cmex2equilibriumLatentCS

SOLVE_ARGS = dict(mu0=1.0, maxIter=30, saveIter=-1, addEye2Hessian=(1e-9, 1e-9))


def matrixsaddle(N1=50, N2=20, classname="tmp_matrixsaddle", **options):
    A1 = tc.Tvariable("A1", (N1, N2))
    u = tc.Tvariable("u", (N1,))
    d = tc.Tvariable("d", (N2,))
    J1 = u * A1 * d
    return cmex2equilibriumLatentCS(classname=classname, P1objective=J1, P2objective=-J1,
                                    P1optimizationVariables=[u], P2optimizationVariables=[d],
                                    P1constraints=[tc.tsum(u, 1) == 1, u >= 0], P2constraints=[tc.tsum(d, 1) == 1, d >= 0],
                                    outputExpressions=[u, d, J1], outputNames=["u", "d", "J1"], parameters=[A1], **options)


def matrixsaddle_data(seed, N1=50, N2=20):
    """``thisA1=rand(N1,N2)``, uniform initial policies (TCgames.m:60-66)"""
    rs = np.random.RandomState(seed)
    A1 = rs.random_sample(N1 * N2).reshape((N1, N2), order="F")
    return dict(A1=A1, u=np.ones(N1) / N1, d=np.ones(N2) / N2)


def qsaddle(N=200, classname="tmp_qsaddle", **options):
    u = tc.Tvariable("u", (N,))
    d = tc.Tvariable("d", (N,))
    Ju = tc.norm2(u + d - 2) - 10 * tc.norm2(d)
    return cmex2equilibriumLatentCS(classname=classname, P1objective=Ju, P2objective=-Ju,
                                    P1optimizationVariables=[u], P2optimizationVariables=[d],
                                    P1constraints=[u >= -1, u <= 1], P2constraints=[d >= -1, d <= 1],
                                    outputExpressions=[u, d, Ju], outputNames=["u", "d", "Ju"], **options)


def qsaddle_data(seed, N=200):
    """TCgames.m:134-137 starts from .5*ones; the sweep perturbs the start inside the box"""
    rs = np.random.RandomState(seed)
    return dict(u=.5 + .4 * (rs.random_sample(N) - .5), d=.5 + .4 * (rs.random_sample(N) - .5))


def qsaddlestate(N=200, classname="tmp_qsaddlestate", **options):
    u = tc.Tvariable("u", (N,))
    d = tc.Tvariable("d", (N,))
    x = tc.Tvariable("x", (N,))
    Ju = tc.norm2(x - 2) - 10 * tc.norm2(d)
    Jd = tc.norm2(x - 2) - 10 * tc.norm2(d)
    return cmex2equilibriumLatentCS(classname=classname, P1objective=Ju, P2objective=-Jd,
                                    P1optimizationVariables=[u], P2optimizationVariables=[d], latentVariables=[x],
                                    P1constraints=[u >= -1, u <= 1], P2constraints=[d >= -1, d <= 1],
                                    latentConstraints=[x == u + d],
                                    outputExpressions=[u, d, x, Ju, Jd], outputNames=["u", "d", "x", "Ju", "Jd"], **options)


def qsaddlestate_data(seed, N=200):
    rs = np.random.RandomState(seed)
    u = .2 * (rs.random_sample(N) - .5)
    d = .2 * (rs.random_sample(N) - .5)
    return dict(u=u, d=d, x=u + d + .1 * (rs.random_sample(N) - .5))


# test-size instances of the three games: name -> (builder, data generator, builder kwargs)
GAMES = {
    "ms": (matrixsaddle, matrixsaddle_data, dict(N1=12, N2=7, classname="tmp_matrixsaddle_12_7")),
    "qs": (qsaddle, qsaddle_data, dict(N=10, classname="tmp_qsaddle_10")),
    "qss": (qsaddlestate, qsaddlestate_data, dict(N=10, classname="tmp_qsaddlestate_10")),
}


def game_data(name, seed):
    mk, data, kw = GAMES[name]
    return data(seed, **{k: v for k, v in kw.items() if k != "classname"})
