"""GPU box: time prebuilt sls class libraries by tag:  python scripts/_time_tags.py w8x_rb150 w8x_rb104 ..."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import __graft_entry__ as g
from tenscalc_amd.problems.sls import SOLVE_ARGS
from tenscalc_amd.runtime import LAYOUT_BATCH_MINOR, HipSolver
B = 65536
gen = torch.Generator(device="cuda").manual_seed(1234)
A = torch.rand((800, B), generator=gen, device="cuda", dtype=torch.float64)
b = torch.rand((100, B), generator=gen, device="cuda", dtype=torch.float64)
x0 = .02 * torch.rand((8, B), generator=gen, device="cuda", dtype=torch.float64)
ref = None
for tag in sys.argv[1:]:
    cls = "tmp_minslsc2h" if tag.startswith("w8s") else "tmp_minslsc2"
    so = os.path.join(g.GENERATED, "@" + cls, "lib%s_%s.so" % (cls, tag))
    obj = HipSolver(so, B=B)
    for nm, t in (("A", A), ("b", b), ("x", x0)):
        obj.set_device(nm, t, LAYOUT_BATCH_MINOR)
    obj.synchronize()
    ts = []
    for _ in range(8):
        st, it, tm = obj.solve(**SOLVE_ARGS)
        ts.append(tm)
    J, x = obj.getOutputs()
    if ref is None:
        ref = (st.copy(), it.copy(), x.copy())
    same = np.array_equal(st, ref[0]) and np.array_equal(it, ref[1]) and np.array_equal(x, ref[2])
    print(tag, "%.3f ms" % (np.median(ts) * 1e3), "solved", float((st == 0).mean()), "same as first:", same, flush=True)
    obj.close()
