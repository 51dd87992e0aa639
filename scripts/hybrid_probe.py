"""Diagnostic (GPU box): do the tile machine (HBM-bound, no LDS) and the LDS walk (LDS / latency-bound, no HBM traffic) overlap
when they run l1l2 side by side on the same CUs?  Two handles, two class libraries, two streams: A instances on the tile machine,
the rest of 16 384 on the LDS walk, both launched, both waited for; against each alone and against the tile machine with all.
   python scripts/hybrid_probe.py <tile library> <lds library> [B_lds ...]"""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import numpy as np
import torch

import __graft_entry__ as g
from tenscalc_amd.problems.l1l2 import SOLVE_ARGS, l1l2, l1l2_data
from tenscalc_amd.runtime import HipSolver

tile_so, lds_so = sys.argv[1], sys.argv[2]
shares = [int(x) for x in sys.argv[3:]] or [2048, 4096]
spec = l1l2(folder=g.GENERATED)
names = [n for n, _ in spec.parameters + spec.variables]
NB = 16384
ds = [l1l2_data(k) for k in range(256)]
st_ = {n: np.stack([np.asarray(ds[k % 256][n], dtype=float).reshape(-1, order="F") for k in range(NB)]) for n in names}
args = dict(mu0=SOLVE_ARGS["mu0"], maxIter=SOLVE_ARGS["maxIter"])


def handle(so, lo, hi):
    h = HipSolver(so, B=hi - lo)
    for n in names:
        h.set(n, st_[n][lo:hi])
    return h


def timed(hs, reps=3):
    best = None
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for h in hs:
            h.solve_async(**args)
        for h in hs:
            h.synchronize()
        t = time.perf_counter() - t0
        best = t if best is None else min(best, t)
    return best * 1e3


full = handle(tile_so, 0, NB)
print("tile machine, all %d: %.1f ms" % (NB, timed([full])), flush=True)
full.close()
for bl in shares:
    a, b = handle(tile_so, 0, NB - bl), handle(lds_so, NB - bl, NB)
    ta, tb, tab = timed([a]), timed([b]), timed([a, b])
    print("tile %5d alone %.1f ms | lds %5d alone %.1f ms | together %.1f ms (sum %.1f, max %.1f)" % (NB - bl, ta, bl, tb, tab, ta + tb, max(ta, tb)), flush=True)
    a.close(); b.close()
