"""Experiment: what would a second resident wavefront per SIMD buy the wave-cooperative lowering?
sls with n=4 unknowns (N=100 rows) needs 246 registers and 14.5 KB of LDS per wavefront, so two
wavefronts per SIMD fit; padding the LDS array (TC_LDS_PAD_SLOTS) forces one per SIMD with the very
same code.  Run on the GPU box:  [TC_PADS=0,200,350] python scripts/occupancy_probe.py [n]
Measured: n=4: 1 -> 2 wavefronts per SIMD = 0.466 -> 0.298 ms; n=2: 1 -> 1.75 -> 3 = 0.355 -> 0.251
-> 0.193 ms."""
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np
import torch

import __graft_entry__ as g
from tenscalc_amd.problems.sls import SOLVE_ARGS, sls
from tenscalc_amd.runtime import LAYOUT_BATCH_MINOR, HipSolver

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
B = 65536
gen = torch.Generator(device="cuda").manual_seed(7)
A = torch.rand((100 * n, B), generator=gen, device="cuda", dtype=torch.float64)
b = torch.rand((100, B), generator=gen, device="cuda", dtype=torch.float64)
x0 = .02 * torch.rand((n, B), generator=gen, device="cuda", dtype=torch.float64)
for pad in [int(p) for p in os.environ.get("TC_PADS", "0,200,350,700").split(",")]:
    os.environ["TC_LDS_PAD_SLOTS"] = str(pad)
    spec = sls(N=100, n=n, classname="tmp_sls_100_%d" % n, folder=g.GENERATED)
    kw = dict(g.VARIANT_KW["w8"])
    kw["tag"] = "w8p%d" % pad
    so = spec.build(**kw)
    obj = HipSolver(so, B=B)
    for nm, t in (("A", A), ("b", b), ("x", x0)):
        obj.set_device(nm, t, LAYOUT_BATCH_MINOR)
    obj.synchronize()
    ts = []
    for _ in range(5):
        st, it, tm = obj.solve(**SOLVE_ARGS)
        ts.append(tm)
    print(f"n={n} pad={pad}: kernel {np.median(ts)*1e3:.3f} ms, solved {np.mean(st == 0):.3f}, iters {it.mean():.2f}", flush=True)
    obj.close()
