"""How small could the per-instance state be?  Replays the entry-point calls of ONE complete solve (the device Newton loop
compiled for the host, oracle/devloop_glue.cpp, over the oracle's tape interpreter), expands them into the sequence of
instruction evaluations the lazy flags would make, and computes instruction-granularity liveness over it: a value is live from
the evaluation that writes it to its last read before the next write.  The maximum number of simultaneously live values is
what a PERFECT allocator would need -- the lower bound for every LDS-resident lowering (mpc_dcmotor
2 301 doubles against the 4 987 slots of the compiled LDS walk).   python scripts/live_set.py [seed]   (CPU only)"""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.getcwd())
from tenscalc_amd.problems.mpc_dcmotor import mpc_dcmotor, dcmotor_data, SOLVE_ARGS
from tenscalc_amd import instructions as I
from tenscalc_amd.tapeio import save_tape
from oracle.tc_oracle import DeviceLoopOracle
spec = mpc_dcmotor()
T = spec.tape
cls = spec.classname
p = os.path.join(tempfile.mkdtemp(), "t.tape"); save_tape(T, p)
o = DeviceLoopOracle(p, "dcm_live", spec.defines)
d = dcmotor_data(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
for nm, _ in spec.parameters + spec.variables:
    o.set(cls + "_set_" + nm, d[nm])
st, it = o.solve(SOLVE_ARGS["mu0"], SOLVE_ARGS["maxIter"], -1, SOLVE_ARGS["addEye2Hessian"])
tr = o.trace()
print("status", st, "iter", it, "calls", len(tr))
# simulate lazy evaluation -> sequence of events: ("g", group) evaluations, reads by gets/copies, writes by sets/copies
ent = {}
for e in T.sets: ent[e["name"]] = ("set", e)
for e in T.gets: ent[e["name"]] = ("get", e)
for e in T.copies: ent[e["name"]] = ("copy", e)
dirty = np.ones(T.nGroups + 1, dtype=bool)
events = []   # (kind, payload)
# initial external sets
for nm, _ in spec.parameters + spec.variables:
    e = ent[cls + "_set_" + nm][1]
    events.append(("w", e["ins"]))
for name in tr + [cls + "_getOutputs"]:
    kind, e = ent[name]
    if kind == "set":
        events.append(("w", e["ins"])); dirty[e["children"]] = True
    else:
        for g in e["parents"]:
            if dirty[g]:
                events.append(("g", g)); dirty[g] = False
        if kind == "get":
            events.append(("r", np.concatenate([o_["ins"] for o_ in e["outputs"]])))
        else:
            events.append(("r", e["src"])); events.append(("w", e["dst"])); dirty[e["children"]] = True
ng = sum(1 for k, _ in events if k == "g")
print("group evals", ng)
# flatten to instruction-level timeline
ost, opl = T.opStart.tolist(), T.ops.tolist()
isload = T.isload
time = 0
last_write = {}; intervals = []  # (start,end) per value instance
last_read = {}
def close(i):
    if i in last_write:
        intervals.append((last_write[i], last_read.get(i, last_write[i]), i))
nacc = 0
for k, pl in events:
    if k == "g":
        for i in T.groups[pl - 1]:
            i = int(i); time += 1
            for j in opl[ost[i - 1]:ost[i]]:
                if not isload[j - 1]:
                    last_read[j] = time; nacc += 1
            close(i); last_write[i] = time; last_read.pop(i, None); nacc += 1
    elif k == "r":
        time += 1
        for j in pl:
            j = int(j)
            if not isload[j - 1]: last_read[j] = time
    else:
        time += 1
        for i in pl:
            i = int(i); close(i); last_write[i] = time; last_read.pop(i, None)
for i in list(last_write): close(i)
print("timeline length", time, "accesses", nacc, "per iteration", nacc / it)
ev = np.zeros(time + 2, dtype=np.int64)
for s, e_, i in intervals:
    if e_ > s:
        ev[s] += 1; ev[e_] -= 1
live = np.cumsum(ev)
print("max live values (instruction granularity, dead values dropped):", live.max(), "mean", live.mean())
# exclude external parameter sets that are only read by parameter-only groups? report count of set slots
nset = sum(len(e["ins"]) for e in T.sets)
print("set slots", nset, "nSlots", T.nSlots, "loads", int(isload.sum()))
