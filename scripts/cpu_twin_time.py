"""CPU baseline of a BASELINE config: the tape printed as reference-style C (oracle/c_emitter.py) +
the reference's own lib/ipmPD_CSsolver.c, gcc -O1, one core.  python scripts/cpu_twin_time.py l1l2|dcmotor|quad [n]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import __graft_entry__ as g
from oracle.c_emitter import CpuTwin
which = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
if which == "l1l2":
    from tenscalc_amd.problems.l1l2 import SOLVE_ARGS, l1l2 as mk, l1l2_data as data
elif which == "quad":
    from tenscalc_amd.problems.mpc_quadcopter import SOLVE_ARGS, mpc_quadcopter as mk, quadcopter_data as data
else:
    from tenscalc_amd.problems.mpc_dcmotor import SOLVE_ARGS, mpc_dcmotor as mk, dcmotor_data as data
spec = mk(folder=g.GENERATED)
t0 = time.time()
tw = CpuTwin(spec)
print(f"twin built in {time.time()-t0:.1f}s", flush=True)
names = [nm for nm, _ in spec.parameters + spec.variables]
ds = [data(1000 + k) for k in range(n)]
inp = {nm: np.stack([np.asarray(d[nm], dtype=float).reshape(-1, order="F") for d in ds]) for nm in names}
t0 = time.time()
out, st, it = tw.solve_batch(inp, n, mu0=SOLVE_ARGS["mu0"], maxIter=SOLVE_ARGS["maxIter"])
dt = time.time() - t0
print(f"{which}: {n} solves in {dt:.3f}s -> {dt/n*1e3:.3f} ms/solve ({n/dt:.1f} solves/s on 1 core), status0={int((st==0).sum())}, iters {it.min()}..{it.max()}, ms/iter {dt/it.sum()*1e3:.3f}")
