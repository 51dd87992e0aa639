"""Tuning aid: one class library under several emitter keyword settings -- built in parallel (CPU container), timed and
compared bit for bit with the first (GPU box).
   python scripts/kw_sweep.py build|run <sls|dcmotor|l1l2|quadcopter> <variant> <B> "k=v,k=v" "k=v" ...     ("" = the variant as it is)"""
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np

import __graft_entry__ as g

mode, which, variant, B = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
sets = sys.argv[5:]
if which == "sls":
    from tenscalc_amd.problems.sls import SOLVE_ARGS
    mk = lambda: g.sls_spec(variant)      # noqa: E731
elif which == "l1l2":
    from tenscalc_amd.problems.l1l2 import SOLVE_ARGS, l1l2, l1l2_data as data
    mk = lambda: l1l2(folder=g.GENERATED)      # noqa: E731
elif which == "quadcopter":
    from tenscalc_amd.problems.mpc_quadcopter import SOLVE_ARGS, mpc_quadcopter, quadcopter_data as data
    mk = lambda: mpc_quadcopter(folder=g.GENERATED)      # noqa: E731
else:
    from tenscalc_amd.problems.mpc_dcmotor import SOLVE_ARGS, dcmotor_data as data, mpc_dcmotor
    mk = lambda: mpc_dcmotor(folder=g.GENERATED)      # noqa: E731


def build(st):
    kw = dict(g.VARIANT_KW[variant])
    for kv in [x for x in st.split(",") if x]:
        k, v = kv.split("=")
        if k == "E":          # E=NAME:VALUE -> environment variable of the emitter (hashed into the build key)
            os.environ[v.split(":")[0]] = v.split(":")[1]
            continue
        if k == "F":          # F=-amdgpu-sched-strategy:max-ilp -> -mllvm -amdgpu-sched-strategy=max-ilp
            kw["extra_flags"] = tuple(kw.get("extra_flags", ())) + ("-mllvm", v.replace(":", "="))
            continue
        if k == "T":          # T=none -> without the variant's optional flags; T=-amdgpu-sched-strategy:max-ilp -> that one instead
            kw["try_flags"] = () if v == "none" else ("-mllvm", v.replace(":", "="))
            continue
        if k == "D":          # D=NAME:VALUE -> -DNAME=VALUE (diagnostic builds, e.g. D=TCL_DUP_CHAIN:2)
            kw["extra_flags"] = tuple(kw.get("extra_flags", ())) + ("-D" + v.replace(":", "="),)
            continue
        kw[k] = int(v) if v.lstrip('-').isdigit() else v
    kw["tag"] = variant + "_k" + st.replace("=", "").replace(",", "_").replace(":", "")
    try:
        return st, mk().build(**kw)
    finally:
        for kv in [x for x in st.split(",") if x.startswith("E=")]:
            os.environ.pop(kv[2:].split(":")[0], None)


if mode == "build":
    import multiprocessing as mp
    with mp.get_context("fork").Pool(4) as pool:
        for st, so in pool.imap_unordered(build, sets):
            print(repr(st), so, flush=True)
else:
    from tenscalc_amd.runtime import LAYOUT_BATCH_MINOR, HipSolver
    spec = mk()
    names = [n for n, _ in spec.parameters + spec.variables]
    if which == "sls":
        import torch
        gen = torch.Generator(device="cuda").manual_seed(1234)
        dev = {"A": torch.rand((800, B), generator=gen, device="cuda", dtype=torch.float64),
               "b": torch.rand((100, B), generator=gen, device="cuda", dtype=torch.float64),
               "x": .02 * torch.rand((8, B), generator=gen, device="cuda", dtype=torch.float64)}
    else:
        ds = [data(k) for k in range(min(B, 256))]
        host = {n: np.stack([np.asarray(ds[k % len(ds)][n], dtype=float).reshape(-1, order="F") for k in range(B)]) for n in names}
    ref = None
    for rep in range(2):
        for st in sets:
            _, so = build(st)
            obj = HipSolver(so, B=B)
            for n in names:
                if which == "sls":
                    obj.set_device(n, dev[n], LAYOUT_BATCH_MINOR)
                else:
                    obj.set(n, host[n])
            obj.synchronize()
            ts = []
            for _ in range(int(os.environ.get('TC_SWEEP_REPS', '10'))):
                st_, it, tm = obj.solve(**SOLVE_ARGS)
                ts.append(tm)
            out = np.concatenate([o.reshape(B, -1) for o in obj.getOutputs()], axis=1)
            obj.close()
            if ref is None:
                ref = (st_, it, out)
            same = np.array_equal(st_, ref[0]) and np.array_equal(it, ref[1]) and np.array_equal(out, ref[2], equal_nan=True)
            print("%s %s B=%d %-24r min %.4f median %.4f ms  bitwise equal to the first: %s" % (which, variant, B, st, min(ts) * 1e3, np.median(ts) * 1e3, same), flush=True)
