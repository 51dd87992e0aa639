"""``python -m tenscalc_amd.tools.build_class <cls>.tape [--lanes P] [--strict] [--out DIR]``

Builds the HIP class library of a solver from a tape file alone (tenscalc_amd/tapeio.py format):
the hand-off point for a host that produces tapes itself -- e.g. a MATLAB TensCalc installation
exporting its csparse object.  Without the producer's tensor-node table
only term loops are rolled; everything else is emitted instruction by instruction.
"""
from __future__ import annotations

import argparse
import os

import numpy as np

from ..cmex2optimizeCS import SolverSpec
from ..csparse import Tape
from ..tapeio import load_records


def tape_from_records(r):
    """inverse of tapeio.tape_records()"""
    T = Tape()
    T.types = r["types"].astype(np.int32)
    T.n = int(T.types.size)
    T.opStart, T.ops, T.parStart, T.pars = r["opStart"], r["ops"], r["parStart"], r["pars"]
    T.group, T.slot = r["group"], r["slot"]
    T.nSlots, T.nGroups = int(r["nSlots"][0]), int(r["nGroups"][0])
    gs = r["grpStart"]
    T.groups = [r["grpInstr"][gs[g]:gs[g + 1]] for g in range(T.nGroups)]
    names = r["defines_names"].split("\n") if r["defines_names"] else []
    T.defines = {k: (int(v) if float(v).is_integer() and k not in _DOUBLES else float(v))
                 for k, v in zip(names, r["defines_values"])}
    T.isset = T.types == 1
    T.isload = T.types == 2
    T.ldl_stats = []
    T.sets, T.gets, T.copies = [], [], []
    enames = r["entry_names"].split("\n")
    for k, (nm, kind) in enumerate(zip(enames, r["entry_kind"])):
        if kind == 0:
            T.sets.append(dict(name=nm, varname=r[f"e{k}_outnames"], size=tuple(int(x) for x in r[f"e{k}_dims"]),
                               ins=r[f"e{k}_a"], external=bool(r[f"e{k}_ext"][0]),
                               children=[int(x) for x in r.get(f"e{k}_children", [])]))
        elif kind == 1:
            b = r[f"e{k}_b"]
            dims = list(r[f"e{k}_dims"])
            onames = r[f"e{k}_outnames"].split("\n")
            outs, q = [], 0
            for o in range(len(b) - 1):
                nd = int(dims[q])
                outs.append(dict(name=onames[o], size=tuple(int(x) for x in dims[q + 1:q + 1 + nd]),
                                 ins=r[f"e{k}_a"][b[o]:b[o + 1]]))
                q += 1 + nd
            T.gets.append(dict(name=nm, outputs=outs, parents=[int(x) for x in r.get(f"e{k}_parents", [])]))
        else:
            T.copies.append(dict(name=nm, dst=r[f"e{k}_a"], src=r[f"e{k}_b"],
                                 parents=[int(x) for x in r.get(f"e{k}_parents", [])],
                                 children=[int(x) for x in r.get(f"e{k}_children", [])]))
    used = np.zeros(T.n + 1, dtype=bool)
    used[1:] = (T.group[1:] > 0) | ((T.slot[1:] >= 0) & T.isload)
    T.used = used
    T.flops = T._count_flops()
    return T


_DOUBLES = {"gradTolerance", "equalTolerance", "desiredDualityGap", "addEye2HessianUtolerance", "alphaMin",
            "alphaMax", "muFactorAggressive", "muFactorConservative"}


def spec_from_tape(path, folder=None):
    r = load_records(path)
    T = tape_from_records(r)
    get = [e for e in T.gets if e["name"].endswith("_getOutputs")]
    if len(get) != 1:
        raise ValueError("tape has no unique <cls>_getOutputs entry point")
    cls = get[0]["name"][:-len("_getOutputs")]
    ext = [e for e in T.sets if e["external"]]
    spec = SolverSpec(cls, T, T.defines, [(e["varname"], e["size"]) for e in ext], [],
                      [(o["name"], o["size"]) for o in get[0]["outputs"]],
                      folder or os.path.dirname(os.path.abspath(path)))
    return spec


def spec_from_cg(basename, defines=None, folder=None):
    """SolverSpec from computation-graph files <basename>.{cg,cgc,cgio,cgs} -- as our save_cg writes them or as the
    reference's saveScalarized does (cgformat.load_cg; a reference-written file carries no solver #define block:
    pass the dict of cmex2optimizeCS.m:303-331)."""
    from ..cgformat import load_cg
    T = load_cg(basename, defines=defines)
    get = [e for e in T.gets if e["name"].endswith("_getOutputs")]
    if len(get) != 1:
        raise ValueError("computation graph has no unique <cls>_getOutputs gateway")
    cls = get[0]["name"][:-len("_getOutputs")]
    ext = [e for e in T.sets if e["external"]]
    return SolverSpec(cls, T, T.defines, [(e["varname"], e["size"]) for e in ext], [],
                      [(o["name"], o["size"]) for o in get[0]["outputs"]],
                      folder or os.path.dirname(os.path.abspath(basename)))


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("tape")
    ap.add_argument("--lanes", type=int, default=1)
    ap.add_argument("--strict", action="store_true", help="-ffp-contract=off, true divisions")
    ap.add_argument("--out", default=None)
    a = ap.parse_args(argv)
    spec = spec_from_tape(a.tape, a.out)
    kw = dict(fp_contract="off") if a.strict else dict(fp_contract="fast", recip=a.lanes > 1, fast_div=a.lanes > 1)
    so = spec.build(lanes=a.lanes, outdir=a.out, **kw)
    print(so)
    return so


if __name__ == "__main__":
    main()
