"""Diagnostic: where do a solve's cycles go?  Builds the sls class with stats=True (s_memtime
stamps around every dependency-group evaluation) and prints, per group, calls and wall cycles per
instance.  Run on the GPU box:  python scripts/group_cycles.py [w8|strict]"""
import os
import sys

sys.path.insert(0, os.getcwd())
import torch

import __graft_entry__ as g
from tenscalc_amd.problems.sls import SOLVE_ARGS, sls
from tenscalc_amd.runtime import LAYOUT_BATCH_MINOR, HipSolver

var = sys.argv[1] if len(sys.argv) > 1 else "w8"
spec = g.sls_spec(var)
kw = dict(g.VARIANT_KW[var])
kw["tag"] = var + "_stats"
mode = sys.argv[2] if len(sys.argv) > 2 else "all"
kw["tag"] = var + "_stats" + ("p" if mode == "phases" else "")
so = spec.build(stats=("phases" if mode == "phases" else True), **kw)
B = 65536
gen = torch.Generator(device="cuda").manual_seed(1234)
A = torch.rand((800, B), generator=gen, device="cuda", dtype=torch.float64)
b = torch.rand((100, B), generator=gen, device="cuda", dtype=torch.float64)
x0 = .02 * torch.rand((8, B), generator=gen, device="cuda", dtype=torch.float64)
obj = HipSolver(so, B=B)
for nm, t in (("A", A), ("b", b), ("x", x0)):
    obj.set_device(nm, t, LAYOUT_BATCH_MINOR)
obj.synchronize()
for _ in range(3):
    st, it, tm = obj.solve(**SOLVE_ARGS)
s = obj.stats()
G = (s.shape[0] - 10) // 2
tot = s[2 * G].mean()
print(f"{var}: kernel {tm*1e3:.3f} ms, mean cycles/instance-thread {tot:.0f}, iters {it.mean():.2f}, minF trials {s[2*G+1].mean():.1f}")
print("phases [init, exit tests, curvature+direction error, alpha bounds, line search, update, mu, -] cycles per solve:",
      " ".join("%.0f" % s[2 * G + 2 + k].mean() for k in range(8)))
T = spec.tape
rows = []
for k in range(G):
    rows.append((s[k].mean(), s[G + k].mean(), k + 1, len(T.groups[k]), int(T.flops[k + 1].sum())))
rows.sort(reverse=True)
acc = 0
print(" grp  instrs  flops  calls  cycles  cyc/call  share")
for cyc, cnt, gid, ni, fl in rows:
    acc += cyc
    print(f"{gid:4d} {ni:7d} {fl:6d} {cnt:6.1f} {cyc:7.0f} {cyc/max(cnt,1e-9):9.0f} {100*cyc/tot:6.1f}%")
print(f"groups total {100*acc/tot:.1f}% of kernel cycles; remainder = loop control, entry points, prologue/epilogue")
