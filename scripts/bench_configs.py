"""Secondary measurements: BASELINE configs[2..4] on ONE GPU, one JSON line per config in bench.py's
vocabulary (solves/s with inputs resident in HBM, HBM roofline from the algorithmic bytes of SURVEY.md 8d,
CPU twin on one host core beside it).

    python scripts/bench_configs.py l1l2|dcmotor|quad [variant] [B] [reps]

Defaults: l1l2 16384 instances (configs[2]); dcmotor 32768 (configs[3], the whole 8-GPU sweep on one GPU);
quad 512 (configs[4]'s per-GPU share of 4096)."""
import json
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import numpy as np

import __graft_entry__ as g
from tenscalc_amd.runtime import HipSolver

which = sys.argv[1]
if which == "l1l2":
    from tenscalc_amd.problems.l1l2 import SOLVE_ARGS, l1l2 as mk, l1l2_data as data
    B0, var0, ncpu, label = 16384, "interp8", 64, "l1l2estimationCS N=200 (1792x1792 KKT), BASELINE configs[2]"
elif which == "quad":
    from tenscalc_amd.problems.mpc_quadcopter import SOLVE_ARGS, mpc_quadcopter as mk, quadcopter_data as data
    B0, var0, ncpu, label = 512, "interp8", 8, "mpc_quadcopter T=100 (1406x1406 KKT), BASELINE configs[4] per-GPU share"
else:
    from tenscalc_amd.problems.mpc_dcmotor import SOLVE_ARGS, mpc_dcmotor as mk, dcmotor_data as data
    B0, var0, ncpu, label = 32768, "strict", 2000, "mpc_dcmotor T=30 (323x323 KKT), BASELINE configs[3]"
variant = sys.argv[2] if len(sys.argv) > 2 else var0
B = int(sys.argv[3]) if len(sys.argv) > 3 else B0
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
spec = mk(folder=g.GENERATED)
so = spec.build(**g.VARIANT_KW[variant])
names = [n for n, _ in spec.parameters + spec.variables]
nd = min(B, 256)
ds = [data(k) for k in range(nd)]
obj = HipSolver(so, B=B)
stacked = {n: np.stack([np.asarray(ds[k % nd][n], dtype=float).reshape(-1, order="F") for k in range(B)]) for n in names}
for n in names:
    obj.set(n, stacked[n])
ts = []
for _ in range(reps):
    st, it, t = obj.solve(**SOLVE_ARGS)
    ts.append(t)
t = float(np.median(ts))
nin = sum(int(np.prod(s)) if len(s) else 1 for _, s in spec.parameters + spec.variables)
nout = sum(int(np.prod(s)) if len(s) else 1 for _, s in spec.outputs)
algo = 8 * (nin + nout) + 8
got = np.concatenate([a.reshape(B, -1) for a in obj.getOutputs()], axis=1)
res = {"metric": "IPM solves/sec, one MI355X", "value": B / t, "unit": "solves/s", "n_gpus": 1, "ms_per_step": t * 1e3,
       "dtype": "f64", "data": "synthetic", "config": {"workload": label, "batch": B, "variant": variant,
                                                         "solved_fraction": float(np.mean(st == 0)),
                                                         "iter_median": float(np.median(it)), "iter_max": int(it.max())},
       "roofline": {"bound": "hbm", "achieved": algo * B / t / 1e9, "peak": 8000.0, "unit": "GB/s",
                    "frac": algo * B / t / 1e9 / 8000.0, "algorithmic_bytes_per_solve": algo, "traffic": None}}
try:
    n = min(B, ncpu)
    sub = {k: v[:n] for k, v in stacked.items()}
    twin = None
    if which != "quad":
        try:
            from oracle.c_emitter import CpuTwin
            twin = CpuTwin(spec)
        except (FileNotFoundError, OSError):
            twin = None          # not prebuilt in this checkout: time the tape interpreter instead
    if twin is None:
        # the generated-C twin of this tape is a 44 MB source file gcc needs the better part of an hour
        # for: time the oracle's tape interpreter instead (same arithmetic, about 2x slower per instruction)
        from tenscalc_amd.tapeio import save_tape
        from oracle.tc_oracle import Oracle
        save_tape(spec.tape, "/tmp/%s.tape" % which)
        c = spec.classname
        o = Oracle("/tmp/%s.tape" % which)
        t0 = time.perf_counter()
        out, st2, it2 = o.solve_batch({f"{c}_set_{k}": v for k, v in sub.items()}, c + "_getOutputs", n,
                                      mu0=SOLVE_ARGS["mu0"], maxIter=SOLVE_ARGS["maxIter"])
        how = "oracle tape interpreter (oracle/tc_oracle.c)"
    else:
        tw = twin
        t0 = time.perf_counter()
        out, st2, it2 = tw.solve_batch(sub, n, mu0=SOLVE_ARGS["mu0"], maxIter=SOLVE_ARGS["maxIter"])
        how = "generated-C twin + reference ipmPD_CSsolver.c, gcc -O1"
    dt = time.perf_counter() - t0
    res["cpu_baseline"] = {"value": n / dt, "unit": "solves/s", "cores": 1, "kind": "port",
                           "sample": f"first {n} instances, {how}, {dt:.2f} s",
                           "same_status_iter": bool(np.array_equal(st[:n], st2) and np.array_equal(it[:n], it2)),
                           "max_abs_output_diff": float(np.nanmax(np.abs(got[:n] - out)))}
except (FileNotFoundError, OSError) as e:
    res["cpu_baseline"] = None
print(json.dumps(res))
