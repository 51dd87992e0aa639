import sys, os, time
sys.path.insert(0, os.getcwd())
import __graft_entry__ as g
from tenscalc_amd.problems.mpc_quadcopter import mpc_quadcopter
from tenscalc_amd.problems.l1l2 import l1l2
from tenscalc_amd.problems.mpc_dcmotor import mpc_dcmotor
from tenscalc_amd.ldsm_schedule import class_stream
for name, mk in (("dcmotor", lambda: mpc_dcmotor(folder=g.GENERATED)), ("l1l2", lambda: l1l2(folder=g.GENERATED)), ("quad", lambda: mpc_quadcopter(folder=g.GENERATED))):
    spec = mk(); T = spec.tape
    for W in (1, 4):
        t0=time.time()
        r = class_stream(T, spec.classname, spec.defines, W=W)
        print(name, "W", W, {k: r[k] for k in ("nLds","nGlob","nExt","nLocal","nHot","nCold","hot_blocks","cold_blocks")}, "words", len(r["words"]), "headers", len(r["headers"])//8, "%.1fs"%(time.time()-t0), flush=True)
