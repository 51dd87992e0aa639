"""Diagnostic: cycle-stamped timeline of ONE wavefront's solve.  Builds the sls class with stats="trace" (lane 0 of
block TC_TRACE_BLOCK writes (tag, s_memtime) pairs at every dependency-group evaluation and phase boundary of the
Newton loop) and prints the per-iteration timeline and the per-tag totals.
   python scripts/trace_solve.py [variant] [key=value,...]      (GPU box)"""
import collections
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np
import torch

import __graft_entry__ as g
from tenscalc_amd.problems.sls import SOLVE_ARGS
from tenscalc_amd.runtime import LAYOUT_BATCH_MINOR, HipSolver

var = sys.argv[1] if len(sys.argv) > 1 else "w8x"
kw = dict(g.VARIANT_KW[var])
if len(sys.argv) > 2 and sys.argv[2]:
    for kv in sys.argv[2].split(","):
        k, v = kv.split("=")
        kw[k] = eval(v)
kw["tag"] = kw.get("tag", var) + "_trace"
spec = g.sls_spec(var)
so = spec.build(stats="trace", **kw)
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
s = obj.stats().reshape(-1)
ev = []
for k in range(0, len(s) - 1, 2):
    ev.append((int(s[k]), int(s[k + 1])))
    if int(s[k]) == 1001:
        break
PH = {100: "init", 101: "exit tests", 102: "curv+direrr", 103: "alpha bounds", 104: "line search", 105: "update", 106: "mu"}
print(f"{var}: kernel {tm*1e3:.3f} ms, {len(ev)} stamps, total {ev[-1][1]-ev[0][1]} cycles, iterations of instance 0: {it[0]}")
grp = collections.defaultdict(lambda: [0, 0])
ph = collections.defaultdict(lambda: [0, 0])
open_g = {}
last_ph = ev[0][1]
lines = []
T = spec.tape
for tag, t in ev:
    if 0 < tag < 100:
        open_g[tag] = t
    elif -100 < tag < 0:
        d = t - open_g.pop(-tag)
        grp[-tag][0] += d
        grp[-tag][1] += 1
        lines.append(f"    g{-tag:<3d} {d:6d}  ({len(T.groups[-tag-1])} instr, {int(T.flops[-tag].sum())} flops)")
    elif 100 <= tag < 108:
        ph[tag][0] += t - last_ph
        ph[tag][1] += 1
        lines.append(f"  == {PH.get(tag, tag)}: {t-last_ph} cycles")
        last_ph = t
tot = ev[-1][1] - ev[0][1]
print("phases:")
for k in sorted(ph):
    print(f"  {PH.get(k,k):14s} {ph[k][0]:8d} cycles {100*ph[k][0]/tot:5.1f}%  {ph[k][1]:3d} x {ph[k][0]/max(1,ph[k][1]):8.0f}")
print("groups (inclusive of nested evaluations):")
for k in sorted(grp, key=lambda q: -grp[q][0]):
    print(f"  g{k:<3d} {grp[k][0]:8d} cycles {100*grp[k][0]/tot:5.1f}%  {grp[k][1]:3d} x {grp[k][0]/grp[k][1]:8.0f}   {len(T.groups[k-1])} instr {int(T.flops[k].sum())} flops")
print("timeline (first 3 passes of the loop):")
n = 0
for ln in lines:
    print(ln)
    if ln.startswith("  == mu"):
        n += 1
        if n == 3:
            break
