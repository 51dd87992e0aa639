"""Timing sweep over prebuilt sls class-library variants (by tag).  GPU box: python scripts/variant_sweep.py w8 strict"""
import sys, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np
import __graft_entry__ as g
from tenscalc_amd.problems.sls import SOLVE_ARGS
from tenscalc_amd.runtime import LAYOUT_BATCH_MINOR, HipSolver
B=65536
gen = torch.Generator(device='cuda').manual_seed(1234)
A = torch.rand((800, B), generator=gen, device='cuda', dtype=torch.float64)
b = torch.rand((100, B), generator=gen, device='cuda', dtype=torch.float64)
x0 = .02 * torch.rand((8, B), generator=gen, device='cuda', dtype=torch.float64)
d=os.path.join(g.GENERATED,'@tmp_minslsc2')
for tag in sys.argv[1:]:
    so=os.path.join(d,'libtmp_minslsc2_%s.so'%tag)
    obj=HipSolver(so,B=B)
    for nm,t in (("A",A),("b",b),("x",x0)): obj.set_device(nm,t,LAYOUT_BATCH_MINOR)
    obj.synchronize()
    ts=[]
    for k in range(8):
        st,it,t=obj.solve(**SOLVE_ARGS); ts.append(t)
    print(tag, 'ms', np.median(ts)*1e3, 'Msolves/s', B/np.median(ts)/1e6, 'ok', (st==0).mean(), flush=True)
    obj.close()
