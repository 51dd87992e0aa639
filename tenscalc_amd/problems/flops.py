"""The reference's flop-count example /root/reference/examples/flops.m:44-62: minimise ``J(x) = x*A*x - b*x`` over
``x`` in R^N, no constraints, parameters ``{A, b}``, outputs ``{J, x}``; one Newton step solves it when A+A' is positive
definite (the file's header).  It is the one problem for which the reference holds numbers of its own for this path:
``flops.m:29-42`` tabulates, for N = 30 .. 300, the operations one ``set A / set b / set x / solve`` sequence executes
(read from the generated library's profile, ``@csparse/writeCprofiling.c:16-22``) and ``code.statistics.sizeScratchbook``.

``FLOPS_PER_SOLVE`` / ``SCRATCHBOOK`` are that table (columns 2 and 5), data, not code.  The flop column was printed by
``fprintf('%10d')`` of a non-integer (total / 1000 repeats), so MATLAB wrote it with seven significant digits:
five entries are exact integers, the last (1.003775e+07) is known to +-5.

What the table pins (tests/test_reference_flops_table.py): tape construction (x*A*x through tprod / gradient / hessian,
nested sums flattened), the scalarizers (tprod, plus, ldl, mldivide, norminf), the CSE of symmetric Hessian entries, the
dependency groups and their lazy evaluation, the Newton loop's call sequence and the per-opcode flop accounting of
``writeCinstructionsC`` -- the product of all of them is one integer per N.  The table was produced by a solver that
never called ``getCurvature__`` / ``getDirectionError__``: it is reproduced EXACTLY with ``adjustAddEye2Hessian=False``
(``ipmPD_CSsolver.c:393-548`` compiled out); today's default (``parameters4optimize.m:116-117``: true) executes
2N^2 + 7N - 3 operations more per solve -- one Hessian-vector product, the curvature's dot product and the direction
error's residual and norm."""
from __future__ import annotations

import numpy as np

from .. import tcalculus as tc
from ..cmex2optimizeCS import cmex2optimizeCS

SOLVE_ARGS = dict(mu0=1.0, maxIter=30, saveIter=-1, addEye2Hessian=(1e-9, 1e-9))       # flops.m:76-79, template default

# examples/flops.m:29-33 (-O0) and :39-42 (-O1): N -> flops per solve (7 significant digits), scratchbook size [doubles]
FLOPS_PER_SOLVE = {30: 1.962100e+04, 60: 1.139460e+05, 100: 4.492460e+05, 150: 1.385121e+06, 200: 3.128496e+06,
                   300: 1.003775e+07}
SCRATCHBOOK = {30: 2138, 60: 7868, 100: 21108, 150: 46658, 200: 82208, 300: 183308}


def flops_problem(N=30, classname=None, **options):
    """``Tvariable x [N]; Tvariable A [N,N]; Tvariable b [N]; J=x*A*x-b*x`` (flops.m:44-48) through cmex2optimizeCS with
    ``solverVerboseLevel=1`` (flops.m:21,59)"""
    x = tc.Tvariable("x", (N,))
    A = tc.Tvariable("A", (N, N))
    b = tc.Tvariable("b", (N,))
    J = x * A * x - b * x
    options.setdefault("solverVerboseLevel", 1)
    return cmex2optimizeCS(classname=classname or "tmp_flops%d" % N, objective=J, optimizationVariables=[x], constraints=[],
                           outputExpressions=[J, x], outputNames=["J", "x"], parameters=[A, b], **options)


def flops_data(seed, N):
    """flops.m:67-68 draws ``A=randn(N,N)`` although its header asks for a positive definite matrix; with an indefinite
    A+A' the solver takes the negative-curvature branch and factors twice, which the table's N^3/3 rules out.  Here
    A = M*M'/N + I + (.25/sqrt(N))*randn (A+A' positive definite, A not symmetric), b = randn; x0 = ones (flops.m:75)."""
    rs = np.random.RandomState(seed)
    M = rs.randn(N, N)
    A = M @ M.T / N + np.eye(N) + 0.25 / np.sqrt(N) * rs.randn(N, N)
    b = rs.randn(N)
    return A, b, np.ones(N)


def flops_batch(B, seed0=0, N=30):
    A = np.empty((B, N, N))
    b = np.empty((B, N))
    x0 = np.empty((B, N))
    for k in range(B):
        A[k], b[k], x0[k] = flops_data(seed0 + k, N)
    return A, b, x0
