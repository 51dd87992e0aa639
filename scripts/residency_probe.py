"""Diagnostic: what would more residents per CU buy the LDS machine on mpc_dcmotor?  A SHORTER horizon (T=22: 3 746 LDS slots
instead of 4 987) fits five workgroups per CU with room to spare; padding its staging area (which does not change its step count) to the size
of the T=30 class brings it back to four.  Same stream, same work per instance: the ratio of the two batch times is what
the fifth resident is worth.
   python scripts/residency_probe.py build          (CPU container)
   python scripts/residency_probe.py run [B]        (GPU box)"""
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np

import __graft_entry__ as g
from tenscalc_amd.problems.mpc_dcmotor import SOLVE_ARGS, dcmotor_data, mpc_dcmotor

T_ = int(os.environ.get("TC_PROBE_T", "22"))
PAD = int(os.environ.get("TC_PROBE_PAD", "1250"))        # staging slots added to reach the LDS footprint of the T=30 class
VARIANTS = {      # name -> (kw overrides, emitter environment)
    "5 residents (stage 512, depth 8, 3 waves/SIMD)": (dict(lds_stage=512, min_waves=3, tag="res5"), {"TC_LDSC_DEPTH": "8"}),
    "4 residents (stage padded, depth 8, 3 waves/SIMD)": (dict(lds_stage=512 + PAD, min_waves=3, tag="res4d8"), {"TC_LDSC_DEPTH": "8"}),
    "4 residents (stage padded, depth 12, 2 waves/SIMD)": (dict(lds_stage=512 + PAD, min_waves=2, tag="res4d12"), {}),
}


def build(name):
    kw, env = VARIANTS[name]
    os.environ.update(env)
    try:
        spec = mpc_dcmotor(T=T_, classname="tmp_dcm%d" % T_, folder=g.GENERATED)
        return spec, spec.build(**dict(g.VARIANT_KW["ldscs"], **kw))
    finally:
        for k in env:
            os.environ.pop(k, None)


if sys.argv[1] == "build":
    from tenscalc_amd import ldsm_schedule as lsch
    for name in VARIANTS:
        spec, so = build(name)
        kw = dict(g.VARIANT_KW["ldscs"], **VARIANTS[name][0])
        cs = lsch.class_stream(lsch.fused_tape(spec.tape, spec.defines), spec.classname, spec.defines, W=kw["waves"],
                               lds_slots=kw["lds_slots"], stage=kw["lds_stage"], split=kw["lds_split_min"])
        print("%-62s nLds %d -> %d workgroups per CU by LDS; cold %d; %s" % (name, cs["nLds"], 160 * 1024 // (8 * cs["nLds"]), cs["nCold"], so))
    sys.exit(0)

from tenscalc_amd.runtime import HipSolver
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
ds = [dcmotor_data(k, T=T_) for k in range(256)]
ref = None
for rep in range(2):
    for name in VARIANTS:
        spec, so = build(name)
        names = [n for n, _ in spec.parameters + spec.variables]
        obj = HipSolver(so, B=B)
        for n in names:
            obj.set(n, np.stack([np.asarray(ds[k % 256][n], dtype=float).reshape(-1, order="F") for k in range(B)]))
        ts = []
        for _ in range(5):
            st, it, tm = obj.solve(**SOLVE_ARGS)
            ts.append(tm)
        out = np.concatenate([o.reshape(B, -1) for o in obj.getOutputs()], axis=1)
        obj.close()
        if ref is None:
            ref = (st, it, out)
        same = np.array_equal(st, ref[0]) and np.array_equal(it, ref[1]) and np.array_equal(out, ref[2])
        print("T=%d B=%d %-62s median %.3f ms  bitwise equal to the first: %s" % (T_, B, name, np.median(ts) * 1e3, same), flush=True)
