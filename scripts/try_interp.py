"""GPU box: parity + timing of body='interp' class libraries against the oracle (development aid)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import __graft_entry__ as g
from tenscalc_amd.runtime import HipSolver
from tenscalc_amd.tapeio import save_tape
from oracle.tc_oracle import Oracle

def run(name, spec, data_fn, args, lib, Bs, Bcheck):
    names = [n for n, _ in spec.parameters + spec.variables]
    c = spec.classname
    tape = f"/tmp/{name}.tape"
    save_tape(spec.tape, tape)
    for B in Bs:
        data = [data_fn(1000 + (k % 256)) for k in range(min(B, 256))]
        stacked = {n: np.stack([np.asarray(data[k % len(data)][n], dtype=float).reshape(-1, order="F") for k in range(B)]) for n in names}
        obj = HipSolver(lib, B)
        for n in names:
            obj.set(n, stacked[n])
        st, it, t = obj.solve(**args)
        st, it, t = obj.solve(**args)
        outs = obj.getOutputs()
        got = np.concatenate([a.reshape(B, -1) for a in outs], axis=1)
        print(f"{name} B={B}: kernel {t*1e3:.2f} ms  {B/t:.0f} solves/s  status0={int((st==0).sum())} iters {it.min()}..{it.max()}", flush=True)
        if B == Bcheck:
            t0 = time.time()
            ref, rst, rit = Oracle(tape).solve_batch({f"{c}_set_{n}": stacked[n] for n in names}, c + "_getOutputs", B,
                                                     mu0=args["mu0"], maxIter=args["maxIter"])
            print(f"   oracle {time.time()-t0:.1f}s; status eq {np.array_equal(st, rst)} iter eq {np.array_equal(it, rit)} "
                  f"outputs bit-equal {np.array_equal(got, ref)} max|diff| {np.nanmax(np.abs(got-ref)):.3g}", flush=True)
        del obj

import os
VAR = os.environ.get("TC_VARIANT", "interp")
which = sys.argv[1:] or ["sls", "dcmotor", "l1l2"]
libs = {t: (b, kw) for t, b, kw in g.solver_variants()}
if "sls" in which:
    from tenscalc_amd.problems.sls import sls, SOLVE_ARGS
    spec = sls(folder=g.GENERATED)
    rs = lambda s: np.random.RandomState(s)
    def d(s):
        r = rs(s); A = r.random_sample((100, 8)); b = r.random_sample(100); x = .02 * r.random_sample(8)
        return dict(A=A, b=b, x=x)
    run("sls", spec, d, SOLVE_ARGS, spec.build(**g.VARIANT_KW[VAR]), [64, 65536], 64)
if "dcmotor" in which:
    from tenscalc_amd.problems.mpc_dcmotor import SOLVE_ARGS, dcmotor_data, mpc_dcmotor
    spec = mpc_dcmotor(folder=g.GENERATED)
    run("dcmotor", spec, dcmotor_data, SOLVE_ARGS, spec.build(**g.VARIANT_KW[VAR]), [64, 4096, 32768], 64)
if "l1l2" in which:
    from tenscalc_amd.problems.l1l2 import SOLVE_ARGS, l1l2, l1l2_data
    spec = l1l2(folder=g.GENERATED)
    run("l1l2", spec, l1l2_data, SOLVE_ARGS, spec.build(**g.VARIANT_KW[VAR]), [8, 2048, 16384], 8)
if "quad" in which:
    from tenscalc_amd.problems.mpc_quadcopter import SOLVE_ARGS, mpc_quadcopter, quadcopter_data
    spec = mpc_quadcopter(folder=g.GENERATED)
    run("quad", spec, quadcopter_data, SOLVE_ARGS, spec.build(**g.VARIANT_KW[VAR]), [4, 512], 4)
