"""CPU: every compile-time branch of the Newton-loop template that cmex2optimizeCS can select
(setAddEye2Hessian, adjustAddEye2Hessian, useInertia, coupledAlphas, scaleInequalities, scaleCost,
nF==0, nG>0) -- restated loop vs the reference's lib/ipmPD_CSsolver.c compiled with the same
#define block, bit-for-bit, on a small problem."""
import numpy as np
import pytest

import tenscalc_amd.tcalculus as tc
from tenscalc_amd.cmex2optimizeCS import cmex2optimizeCS

OPTIONS = [
    dict(),
    dict(useInertia=True),
    dict(coupledAlphas=False),
    dict(scaleInequalities=False),
    dict(addEye2Hessian=False, adjustAddEye2Hessian=False),
    dict(adjustAddEye2Hessian=False),
    dict(scaleCost=1.0),
    dict(useInertia=True, coupledAlphas=False, scaleCost=10.0, desiredDualityGap=1e-7, gradTolerance=1e-6),
    dict(muFactorAggressive=.5, muFactorConservative=.8, alphaMin=1e-5, alphaMax=.9),
]


def _problem(kind, **opt):
    N, n = 12, 4
    A = tc.Tvariable("A", (N, n)); b = tc.Tvariable("b", (N,)); x = tc.Tvariable("x", (n,))
    J = tc.norm2(A * x - b) / N
    if kind == "box":
        cons = [x >= 0, x <= .05]
    elif kind == "eq":                       # equality + inequality: sum(x)==.1, x>=0
        cons = [tc.tsum(x) == .1, x >= 0]
    elif kind == "eqonly":                   # nF==0 branch
        cons = [tc.tsum(x) == .1]
    else:                                    # unconstrained: nF==0, nG==0
        cons = []
    return cmex2optimizeCS(classname="tmp_opt_" + kind, objective=J, optimizationVariables=[x], constraints=cons,
                           outputExpressions=[J, x], outputNames=["J", "x"], parameters=[A, b], **opt)


@pytest.mark.parametrize("kind", ["box", "eq", "eqonly", "free"])
@pytest.mark.parametrize("opt", OPTIONS, ids=[str(k) for k in range(len(OPTIONS))])
def test_loop_branches_bitwise_vs_reference(native_built, tmp_path, kind, opt):
    from tenscalc_amd.tapeio import save_tape
    from oracle.tc_oracle import Oracle, ReferenceLoopOracle
    if kind in ("eqonly", "free") and ("scaleInequalities" in opt or "coupledAlphas" in opt):
        pytest.skip("option only matters with inequality constraints")
    spec = _problem(kind, **opt)
    tape = str(tmp_path / "p.tape")
    save_tape(spec.tape, tape)
    tag = "opt_%s_%d" % (kind, OPTIONS.index(opt))
    try:
        ro = ReferenceLoopOracle(tape, tag, spec.defines)
    except (FileNotFoundError, OSError):
        pytest.skip("oracle/_ref not available")
    o = Oracle(tape)
    c = spec.classname
    rs = np.random.RandomState(3)
    nconv = 0
    for trial in range(6):
        A, b, x0 = rs.random_sample((12, 4)), rs.random_sample(12), .02 * rs.random_sample(4) + .005
        r = []
        for q in (o, ro):
            q.set(c + "_set_A", A); q.set(c + "_set_b", b); q.set(c + "_set_x", x0)
            st, it = q.solve(1.0 if trial % 2 else 1e-2, 40, -1, (1e-9, 1e-9))
            r.append((st, it, q.get(c + "_getOutputs")))
        assert r[0][:2] == r[1][:2], (kind, opt, r[0][:2], r[1][:2])
        assert np.array_equal(r[0][2], r[1][2], equal_nan=True)
        nconv += r[0][0] == 0
    assert nconv >= 4


def test_allow_save_declares_saveWW_and_writes_the_subscripts_file(native_built, tmp_path):
    """cmex2optimizeCS(allowSave=true): the Newton matrix becomes reachable through saveWW__ and its sparsity structure is
    written in the reference's format next to the class (ipmPD_CStimesLambda.m:283-291, writeCfunctionpergroup.m:905-916)"""
    from tenscalc_amd import typical
    from tenscalc_amd.problems.sls import sls
    spec = sls(N=20, n=4, classname="tmp_sls_save", folder=str(tmp_path), allowSave=True)
    assert spec.tape.has("saveWW__")
    magic, osize, subs = typical.read_subscripts(str(tmp_path / "@tmp_sls_save" / "tmp_sls_save_WW.subscripts"))
    n = 4 + 8                                   # nU + nF
    assert osize == (n, n) and len(subs) == spec.tape.ldl_stats[-1]["nnzA"] and magic > 0
    plain = sls(N=20, n=4, classname="tmp_sls_nosave", folder=str(tmp_path))
    assert not plain.tape.has("saveWW__")
