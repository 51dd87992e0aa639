"""Tuning aid for the LDS-resident back-ends at full batch: residents per CU against per-solve latency.
   python scripts/lds_budget_sweep.py build|run <dcmotor|l1l2|quad> <variant> <B> <order:stage:lds_slots> ...
order = mindeg | mmdK | depthW (scripts/order_ab.py), optional 4th field "fine" (TC_FINE_LIVES), stage = staging slots (TC_LDSM_STAGE), lds_slots = LDS budget in doubles
(160 KB per CU = 20 480 doubles: 5 120 -> 4 instances per CU, 4 096 -> 5, 3 413 -> 6, 2 925 -> 7, 2 560 -> 8)."""
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np

import __graft_entry__ as g

mode, which, variant, B = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
if which == "l1l2":
    from tenscalc_amd.problems.l1l2 import SOLVE_ARGS, l1l2 as mk, l1l2_data as data
elif which == "quad":
    from tenscalc_amd.problems.mpc_quadcopter import SOLVE_ARGS, mpc_quadcopter as mk, quadcopter_data as data
else:
    from tenscalc_amd.problems.mpc_dcmotor import SOLVE_ARGS, mpc_dcmotor as mk, dcmotor_data as data


def order(nm):
    if nm == "mindeg":
        return "mindeg"
    if nm.startswith("mmd"):
        t, _, gate = nm[3:].partition("_")
        return ("mmd", int(t), gate) if gate else ("mmd", int(t))
    return ("depth", int(nm[5:]))


def build(cfg):
    from tenscalc_amd import tcalculus as tc
    o, stage, slots, *rest = cfg.split(":")
    os.environ["TC_LDSM_STAGE"] = stage
    from tenscalc_amd.lowering import Lowering
    Lowering.FINE_LIVES = bool(rest and rest[0] == "fine")       # finer life times of the exported values
    tc.clear()
    spec = mk(folder=g.GENERATED, classname="tmpo_%s_%s" % (which, o), ldlOrder=order(o))
    kw = dict(g.VARIANT_KW[variant], lds_slots=int(slots), tag="%s_s%s_l%s%s" % (variant, stage, slots, "_fine" if rest else ""))
    return cfg, spec, spec.build(**kw)


def build_only(cfg):
    c, _, so = build(cfg)
    return c, so


cfgs = sys.argv[5:]
if mode == "build":
    import multiprocessing as mp
    with mp.get_context("fork").Pool(min(6, len(cfgs))) as pool:
        for cfg, so in pool.imap_unordered(build_only, cfgs):
            print(cfg, so, flush=True)
    sys.exit(0)
if len(cfgs) > 1:          # one process per library: several class libraries in one process share their kernel symbol names
    import subprocess
    seen = {}
    for cfg in cfgs:
        r = subprocess.run([sys.executable, sys.argv[0], "run", which, variant, str(B), cfg], capture_output=True, text=True)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith(which)]
        if not line:
            print(cfg, "FAILED", r.stderr[-500:], flush=True)
            continue
        h = line[-1].rsplit("sha ", 1)[1]
        key = cfg.split(":")[0]
        print(line[-1], "(first of its order)" if key not in seen else ("bitwise equal to the first: %s" % (seen[key] == h)), flush=True)
        seen.setdefault(key, h)
    sys.exit(0)
import hashlib
from tenscalc_amd.runtime import HipSolver
nd = min(B, 512)
ds = [data(k) for k in range(nd)]
ref = {}
for cfg in cfgs:
    _, spec, so = build(cfg)
    names = [n for n, _ in spec.parameters + spec.variables]
    obj = HipSolver(so, B=B)
    for n in names:
        obj.set(n, np.stack([np.asarray(ds[k % nd][n], dtype=float).reshape(-1, order="F") for k in range(B)]))
    ts = []
    for _ in range(4):
        st, it, t = obj.solve(**SOLVE_ARGS)
        ts.append(t)
    outs = np.concatenate([np.asarray(o_).reshape(B, -1) for o_ in obj.getOutputs()], axis=1)
    sha = hashlib.sha256(outs.tobytes() + it.tobytes() + st.tobytes()).hexdigest()[:16]
    print("%s %s B=%d %-22s min %8.3f ms  %.3g solves/s  iters median %d max %d solved %d  sha %s" % (
        which, variant, B, cfg, min(ts) * 1e3, B / min(ts), np.median(it), it.max(), int((st == 0).sum()), sha), flush=True)
    obj.close()
