"""Diagnostic (CPU): how the rows one record of the tile machine touches lie in the tile -- alone or in runs of consecutive rows.
Read with scripts/ubench/tile_rows.hip (the HBM bandwidth of scattered 512-byte rows against runs of them): DESIGN.md 3.4.
   python scripts/interp_row_runs.py"""
import sys, numpy as np, collections
sys.path.insert(0, "/root/repo")
import __graft_entry__ as g
from tenscalc_amd import interp_schedule as isch, ldsm_schedule as lsch
from tenscalc_amd.lowering import Lowering
from tenscalc_amd.problems.l1l2 import l1l2
spec = l1l2(folder=g.GENERATED)
T = lsch.fused_tape(spec.tape, spec.defines)
L = Lowering(T, None, roll=False, lanes=1, spill_all=True)
cs = isch.class_stream(T, L, spec.classname, 8)
w = cs["words"]
tot = al = adj = 0
nrec = 0
runs = collections.Counter()
for rg in cs["granges"]:
    for (s, e) in rg:
        pc = s
        while pc < e:
            if (w[pc] >> 8) == 2 and (w[pc] & 255) == 0:
                pc += w[pc + 6]; continue
            rows, ln = isch._record_rows(w, pc)
            pc += ln
            st = set(rows)
            nrec += 1
            tot += len(st)
            al += sum(1 for r in st if (r ^ 1) in st)
            adj += sum(1 for r in st if (r + 1) in st or (r - 1) in st)
            # run lengths
            srt = sorted(st); k = 0
            while k < len(srt):
                j = k
                while j + 1 < len(srt) and srt[j + 1] == srt[j] + 1: j += 1
                runs[min(j - k + 1, 8)] += j - k + 1
                k = j + 1
print("records", nrec, "distinct rows touched", tot, "with aligned partner %.1f%%" % (100 * al / tot), "with any neighbour %.1f%%" % (100 * adj / tot))
print("rows by run length (8 = 8+):", {k: "%.1f%%" % (100 * v / tot) for k, v in sorted(runs.items())})
print("tile rows", cs["hi"] - cs["lo"])
