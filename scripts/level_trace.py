"""Where a level's cycles go in the compiled walk: a build with stats="levels" stamps (record, clock) whenever
wavefront 0 of block 0 enters a step; this script turns the stamps of ONE solve into a table of step durations by what
the step holds (all wavefronts' records of that step).
   python scripts/level_trace.py build <quadcopter|l1l2|dcmotor> [variant]     (CPU container)
   python scripts/level_trace.py run   <quadcopter|l1l2|dcmotor> [variant]     (GPU box)"""
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.getcwd())
import numpy as np

import __graft_entry__ as g
from tenscalc_amd import instructions as I
from tenscalc_amd import ldsm_schedule as lsch

mode, which = sys.argv[1], sys.argv[2]
variant = sys.argv[3] if len(sys.argv) > 3 else "ldsc4"
if which == "quadcopter":
    from tenscalc_amd.problems.mpc_quadcopter import SOLVE_ARGS, mpc_quadcopter as mk, quadcopter_data as data
elif which == "l1l2":
    from tenscalc_amd.problems.l1l2 import SOLVE_ARGS, l1l2 as mk, l1l2_data as data
else:
    from tenscalc_amd.problems.mpc_dcmotor import SOLVE_ARGS, dcmotor_data as data, mpc_dcmotor as mk
spec = mk(folder=g.GENERATED)
kw = dict(g.VARIANT_KW[variant])
kw["stats"] = "levels"
kw["tag"] = variant + "_levels"
so = spec.build(**kw)
meta = os.path.splitext(so)[0] + ".levels.npz"
if mode == "build":
    T = lsch.fused_tape(spec.tape, spec.defines)
    cs = lsch.class_stream(T, spec.classname, spec.defines, W=int(kw["waves"]), lds_slots=kw.get("lds_slots", 20000),
                           stage=int(kw.get("lds_stage", 0)), split=int(kw.get("lds_split_min", 0)) or True)
    np.savez(meta, headers=np.asarray(cs["headers"], dtype=np.int64).reshape(-1, 8), table=np.asarray(cs["table"], dtype=np.int64))
    print(so, meta)
    sys.exit(0)

from tenscalc_amd.runtime import HipSolver
B = 64
m = np.load(meta)
hd, table = m["headers"], m["table"]
W = int(kw["waves"])
names = [n for n, _ in spec.parameters + spec.variables]
ds = [data(k) for k in range(B)]
obj = HipSolver(so, B=B)
for n in names:
    obj.set(n, np.stack([np.asarray(d[n], dtype=float).reshape(-1, order="F") for d in ds]))
st, it, tm = obj.solve(**SOLVE_ARGS)
raw = obj.stats().reshape(-1)
obj.close()
print("instance 0: status %d, %d iterations; batch %.2f ms" % (st[0], it[0], tm * 1e3))
rec, clk = raw[0::2].astype(np.int64), raw[1::2]
nst = int(np.argmax(clk == 0)) if (clk == 0).any() else len(clk)
rec, clk = rec[:nst], clk[:nst]
print(nst, "stamps")
# the step a record of wavefront 0 opens: the records of every wavefront between that step's start and the next one
opname = {v: k for k, v in vars(I).items() if k.startswith("I_") and isinstance(v, int)}
kindname = {0: "long", 1: "short", 2: "barrier", 3: "fold"}
# per record range and wavefront: list of steps (lists of record indices)
steps_of = {}      # first record of wavefront 0's step -> signature
nrid = len(table) // (2 * W)
for rid in range(nrid):
    per = []
    for k in range(W):
        h0, h1 = int(table[(rid * W + k) * 2]), int(table[(rid * W + k) * 2 + 1])
        cur, out = None, []
        for h in range(h0, h1):
            new = bool(hd[h, 5] & 4)
            if new or cur is None:
                cur = []
                out.append(cur)
            cur.append(h)
        per.append(out)
    n0 = len(per[0])
    for s in range(n0):
        sig = []
        for k in range(W):
            if s < len(per[k]):
                for h in per[k][s]:
                    kind, op = int(hd[h, 0]) >> 8, int(hd[h, 0]) & 255
                    if kind == 2:
                        continue
                    late = (int(hd[h, 5]) >> 3) & 3 if kind == 3 else 0
                    sig.append((kindname[kind], opname.get(op, str(op)), int(hd[h, 1]), int(hd[h, 3]), late, bool(hd[h, 5] & 3)))
        steps_of[per[0][s][0]] = sig
ph, rec = rec & 7, rec >> 3
agg = defaultdict(list)
q = 0
while q < nst - 1:
    if ph[q] != 0:
        q += 1
        continue
    e = q + 1
    while e < nst and ph[e] != 0:
        e += 1
    if e >= nst:
        break
    t = {int(ph[j]): clk[j] for j in range(q, e)}
    sig = steps_of.get(int(rec[q]))
    if sig is not None:
        kinds = sorted({"%s:%s%s%s" % (a, b.replace("I_", ""), "+late" if e_ else "", "+hbm" if f else "") for a, b, c, d, e_, f in sig})
        mx = max([c for a, b, c, d, e_, f in sig if a == "fold"] + [0])
        row = [clk[e] - clk[q], t.get(1, np.nan) - t[0], t.get(2, np.nan) - t.get(1, np.nan), t.get(3, np.nan) - t.get(2, np.nan),
               clk[e] - t.get(3, t.get(2, t[0]))]
        agg[(" ".join(kinds) or "(empty)", min(len(sig), 9), (mx + 7) // 8 * 8)].append(row)
    q = e
tot = sum(r[0] for v in agg.values() for r in v)
nsteps = sum(len(v) for v in agg.values())
print("total stamped cycles %.3g over %d steps: %.0f per step (each phase includes one stamp: two LDS reads, clock, store)" % (tot, nsteps, tot / max(1, nsteps)))
print("%-58s %4s %5s %6s %6s | %6s %6s %6s %6s %6s" % ("step holds (all wavefronts)", "recs", "fold<=", "steps", "share", "total", "gather", "finish", "barrier", "tail"))
for key, v in sorted(agg.items(), key=lambda kv: -sum(r[0] for r in kv[1]))[:40]:
    m_ = np.nanmedian(np.asarray(v, dtype=float), axis=0)
    print("%-58s %4d %5d %6d %5.1f%% | %6.0f %6.0f %6.0f %6.0f %6.0f" % (key[0][:58], key[1], key[2], len(v), 100 * sum(r[0] for r in v) / tot, *m_))
