"""Secondary measurement (BASELINE configs[3]): mpc_dcmotor T=30, one GPU's share of the 32768-instance
sweep (4096 instances), generic one-thread-per-instance lowering.  python scripts/bench_dcmotor.py [B]"""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import numpy as np

import __graft_entry__ as g
from tenscalc_amd.problems.mpc_dcmotor import SOLVE_ARGS, dcmotor_data, mpc_dcmotor
from tenscalc_amd.runtime import HipSolver

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
variant = sys.argv[2] if len(sys.argv) > 2 else "strict"
spec = mpc_dcmotor(folder=g.GENERATED)
if variant == "strict":
    so = spec.build(**g.VARIANT_KW["strict"])
else:                                    # e.g. w64: 64 lanes share one instance, state in LDS
    base, _, extra = variant.partition("_")   # experiments: w32x_s8 (sched_every=8), w32x_inl (everything inlined)
    exact = base.endswith("x")              # e.g. w32x: the bit-exact wave-cooperative lowering
    kw = dict(g.VARIANT_KW["w8x" if exact else "w8"])
    share = base.endswith("s") or exact     # e.g. w64s: LDS slots shared by life time
    kw.update(lanes=int(base[1:].rstrip("sx")), tag=variant, inline_max=32, share=share)
    for tok in [x for x in extra.split("-") if x]:     # e.g. w32x_inl-s8-fw12
        if tok == "inl":
            kw.update(inline_max=1000)
        elif tok[0] == "s":
            kw.update(sched_every=int(tok[1:]))
        elif tok[:2] == "fw":
            kw.update(fw_window=int(tok[2:]))
    so = spec.build(**kw)
data = [dcmotor_data(k) for k in range(B)]
names = [n for n, _ in spec.parameters + spec.variables]
obj = HipSolver(so, B=B)
for n in names:
    obj.set(n, np.stack([np.asarray(d[n], dtype=float).reshape(-1, order="F") for d in data]))
ts = []
for _ in range(5):
    st, it, t = obj.solve(**SOLVE_ARGS)
    ts.append(t)
t = float(np.median(ts))
nin = sum(int(np.prod(s)) if len(s) else 1 for _, s in spec.parameters + spec.variables)
nout = sum(int(np.prod(s)) if len(s) else 1 for _, s in spec.outputs)
algo = 8 * (nin + nout) + 8
print(f"mpc_dcmotor {variant} B={B}: {t*1e3:.2f} ms/batch, {B/t:.3e} solves/s, status0 {np.mean(st==0):.4f}, "
      f"iter median {np.median(it)}, max {it.max()}; algorithmic {algo} B/solve -> {algo*B/t/1e9:.2f} GB/s")
try:
    from oracle.c_emitter import CpuTwin
    tw = CpuTwin(spec)
    n = min(B, 2000)
    stacked = {k: np.stack([np.asarray(d[k], dtype=float).reshape(-1, order="F") for d in data[:n]]) for k in names}
    t0 = time.perf_counter()
    out, st2, it2 = tw.solve_batch(stacked, n, mu0=SOLVE_ARGS["mu0"], maxIter=SOLVE_ARGS["maxIter"])
    dt = time.perf_counter() - t0
    outs = obj.getOutputs()
    got = np.concatenate([a.reshape(B, -1) for a in outs], axis=1)[:n]
    print(f"cpu twin (1 core): {dt/n*1e6:.1f} us/solve = {n/dt:.3e} solves/s; identical status/iter: "
          f"{np.array_equal(st[:n], st2) and np.array_equal(it[:n], it2)}; max |output diff| {np.abs(got-out).max():.3e}, "
          f"|J diff| {np.abs(got[:,0]-out[:,0]).max():.3e}")
except (FileNotFoundError, OSError) as e:
    print("cpu twin unavailable", e)
