"""Generates tests/golden/{dcmotor,l1l2,quadcopter}.npz and tests/golden/sls_trace.npz (run in the development
container, where /root/reference is present).

Expected outputs come from the CPU oracle's tape interpreter driven by the REFERENCE's own Newton loop
(lib/ipmPD_CSsolver.c compiled in place with the problem's -D block: oracle/build.py:build_ref,
ReferenceLoopOracle); the restated loop of oracle/tc_oracle.c must agree bit for bit or generation aborts.
Inputs are reproducible from the seeds (tenscalc_amd.problems.*.<problem>_data), only a checksum is stored.
mpc_quadcopter additionally stores the outputs after ONE Newton iteration (maxIter=1): its tape calls libm pow(),
which differs by an ulp between glibc and the device, so complete solves can only be compared where the iteration
took the same path (tests/test_gpu_interp.py).

sls_trace.npz: for 12 seeds the state after every Newton iteration k = 1..iter (the solve stopped by maxIter=k):
outputs J, x and the loop quantities |grad|_inf, duality gap, min F, min lambda, mu, alphaPrimal."""
import ctypes
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from tenscalc_amd.tapeio import save_tape  # noqa: E402
from oracle.tc_oracle import Oracle, ReferenceLoopOracle  # noqa: E402


def _solve_all(spec, tag, data_fn, seeds, args, extra_maxiter=()):
    tape = "/tmp/golden_%s.tape" % tag
    save_tape(spec.tape, tape)
    c = spec.classname
    o = Oracle(tape)
    ro = ReferenceLoopOracle(tape, tag, spec.defines)
    names = [n for n, _ in spec.parameters + spec.variables]
    out = dict(status=[], iter=[], outputs=[], checksum=[])
    few = {mi: [] for mi in extra_maxiter}
    for seed in seeds:
        d = data_fn(seed)
        res = []
        for q in (o, ro):
            for n in names:
                q.set(f"{c}_set_{n}", d[n])
            st, it = q.solve(args["mu0"], args["maxIter"], -1, args["addEye2Hessian"])
            res.append((st, it, q.get(c + "_getOutputs")))
        if res[0][:2] != res[1][:2] or not np.array_equal(res[0][2], res[1][2], equal_nan=True):
            raise SystemExit(f"{tag} seed {seed}: restated Newton loop disagrees with the reference loop")
        st, it, y = res[1]
        out["status"].append(st); out["iter"].append(it); out["outputs"].append(y)
        out["checksum"].append(sum((k + 1) * float(np.sum(d[n])) for k, n in enumerate(names)))
        for mi in extra_maxiter:
            for n in names:
                ro.set(f"{c}_set_{n}", d[n])
            ro.solve(args["mu0"], mi, -1, args["addEye2Hessian"])
            few[mi].append(ro.get(c + "_getOutputs"))
        print(tag, "seed", seed, "status", st, "iter", it, flush=True)
    kw = {"outputs_maxIter%d" % mi: np.array(v) for mi, v in few.items()}
    np.savez_compressed(os.path.join(HERE, tag + ".npz"), seeds=np.array(list(seeds)),
                        status=np.array(out["status"], dtype=np.int32), iter=np.array(out["iter"], dtype=np.int32),
                        outputs=np.array(out["outputs"]), checksum=np.array(out["checksum"]),
                        output_names=np.array([nm for nm, _ in spec.outputs]),
                        output_numel=np.array([int(np.prod(sz)) if len(sz) else 1 for _, sz in spec.outputs]),
                        newton_loop=np.array("reference ipmPD_CSsolver.c"), **kw)


def sls_trace(nseeds=12):
    from tenscalc_amd.problems.sls import SOLVE_ARGS, sls, sls_data
    spec = sls(folder="/tmp/golden_gen")
    tape = "/tmp/golden_sls.tape"
    save_tape(spec.tape, tape)
    c = spec.classname
    ro = ReferenceLoopOracle(tape, "sls", spec.defines)
    T = spec.tape
    mu_ins = int(T.entry("setMu__")[2]["ins"][0])
    al_ins = int(T.entry("setAlphaPrimal__")[2]["ins"][0])
    val = lambda i: ro.L.tco_value(ro.h, ctypes.c_int64(i))      # noqa: E731
    rows, first, its = [], [], []
    for seed in range(nseeds):
        A, b, x0 = sls_data(seed)

        def run(mi):
            ro.set(c + "_set_A", A); ro.set(c + "_set_b", b); ro.set(c + "_set_x", x0)
            return ro.solve(SOLVE_ARGS["mu0"], mi, -1, SOLVE_ARGS["addEye2Hessian"])
        st, it = run(SOLVE_ARGS["maxIter"])
        assert st == 0
        first.append(len(rows)); its.append(it)
        for k in range(1, it):                    # stopped by maxIter=k: the state after k Newton iterations
            stk, itk = run(k)
            y = ro.get(c + "_getOutputs")
            g = ro.get("getNorminf_Grad__")[0]
            gap, minF, minL = ro.get("getGapMinFMinLambda__")
            rows.append(np.r_[seed, k, stk, itk, y, g, gap, minF, minL, val(mu_ins), val(al_ins)])
    np.savez_compressed(os.path.join(HERE, "sls_trace.npz"), rows=np.array(rows), first_row=np.array(first),
                        iters=np.array(its),
                        columns=np.array(["seed", "maxIter", "status", "iter", "J", "x1", "x2", "x3", "x4", "x5", "x6",
                                          "x7", "x8", "norminf_grad", "gap", "minF", "minLambda", "mu", "alphaPrimal"]),
                        newton_loop=np.array("reference ipmPD_CSsolver.c"))
    print("sls trace:", len(rows), "rows for", nseeds, "seeds")


def main(which=("sls_trace", "dcmotor", "l1l2", "quadcopter")):
    """python tests/golden/make_golden_configs.py [sls_trace] [dcmotor] [l1l2] [quadcopter]   (default: all)"""
    from tenscalc_amd.problems.mpc_dcmotor import SOLVE_ARGS as A1, dcmotor_data, mpc_dcmotor
    from tenscalc_amd.problems.l1l2 import SOLVE_ARGS as A2, l1l2, l1l2_data
    from tenscalc_amd.problems.mpc_quadcopter import SOLVE_ARGS as A3, mpc_quadcopter, quadcopter_data
    if "sls_trace" in which:
        sls_trace()
    if "dcmotor" in which:
        _solve_all(mpc_dcmotor(folder="/tmp/golden_gen"), "dcmotor", dcmotor_data, range(100, 132), A1)
    if "l1l2" in which:
        _solve_all(l1l2(folder="/tmp/golden_gen"), "l1l2", l1l2_data, range(10, 16), A2)
    if "quadcopter" in which:
        _solve_all(mpc_quadcopter(folder="/tmp/golden_gen"), "quadcopter", quadcopter_data, range(20, 26), A3, extra_maxiter=(1,))


if __name__ == "__main__":
    main(tuple(sys.argv[1:]) or ("sls_trace", "dcmotor", "l1l2", "quadcopter"))
