"""Generates tests/golden/sls_N100_n8.npz (run in the development container).

Expected outputs come from the CPU oracle (oracle/tc_oracle.c) whose Newton loop is, for this
file, the reference's own lib/ipmPD_CSsolver.c compiled in place (oracle/_ref, ReferenceLoopOracle)
when /root/reference is available; the restated loop must agree bit-for-bit or generation aborts.
Inputs are reproducible from the seeds (tenscalc_amd.problems.sls.sls_data), so only a checksum of
them is stored.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from tenscalc_amd.problems.sls import SOLVE_ARGS, sls, sls_data  # noqa: E402
from tenscalc_amd.tapeio import save_tape  # noqa: E402
from oracle.tc_oracle import Oracle, ReferenceLoopOracle  # noqa: E402

PROBES = ["getJ__", "getNorminf_Grad__", "getGapMinFMinLambda__", "getCurvature__",
          "getDirectionError__", "getMaxAlphas_s__", "getGrad__"]


def main(nseeds=64):
    spec = sls(folder="/tmp/golden_gen")
    tape = "/tmp/golden_sls.tape"
    save_tape(spec.tape, tape)
    c = spec.classname
    o = Oracle(tape)
    try:
        ro = ReferenceLoopOracle(tape, "sls", spec.defines)
    except FileNotFoundError:
        ro = None
    out = dict(status=[], iter=[], J=[], x=[], checksum=[], loop=[])
    probes = {p: [] for p in PROBES}
    for seed in range(nseeds):
        A, b, x0 = sls_data(seed)
        res = []
        for q in [o] + ([ro] if ro else []):
            q.set(c + "_set_A", A)
            q.set(c + "_set_b", b)
            q.set(c + "_set_x", x0)
            st, it = q.solve(SOLVE_ARGS["mu0"], SOLVE_ARGS["maxIter"], -1, SOLVE_ARGS["addEye2Hessian"])
            res.append((st, it, q.get(c + "_getOutputs")))
        if ro and (res[0][:2] != res[1][:2] or not np.array_equal(res[0][2], res[1][2])):
            raise SystemExit(f"seed {seed}: restated Newton loop disagrees with the reference loop")
        st, it, y = res[-1]
        out["status"].append(st)
        out["iter"].append(it)
        out["J"].append(y[0])
        out["x"].append(y[1:])
        out["checksum"].append(A.sum() + 2 * b.sum() + 3 * x0.sum())
        # tape values right after the solver's initialisation sequence
        o.set(c + "_set_A", A); o.set(c + "_set_b", b); o.set(c + "_set_x", x0)
        o.set("setAddEye2HessianU__", [1e-9]); o.set("setAddEye2HessianEq__", [1e-9])
        o.copy("initPrimal__"); o.copy("scaleIneq__"); o.set("setMu__", [SOLVE_ARGS["mu0"]]); o.copy("initDualIneq__")
        for p in PROBES:
            probes[p].append(o.get(p))
    np.savez(os.path.join(os.path.dirname(os.path.abspath(__file__)), "sls_N100_n8.npz"),
             seeds=np.arange(nseeds), status=np.array(out["status"], dtype=np.int32),
             iter=np.array(out["iter"], dtype=np.int32), J=np.array(out["J"]), x=np.array(out["x"]),
             checksum=np.array(out["checksum"]),
             newton_loop=np.array("reference ipmPD_CSsolver.c" if ro else "oracle restatement"),
             **{"probe_" + p: np.array(v) for p, v in probes.items()})
    print("wrote golden for", nseeds, "seeds; loop:", "reference" if ro else "restated")


if __name__ == "__main__":
    main()
