"""``cmex2compute``-style tapes: a csparse object with only set/get entry points (no solver).

Minimal mirror of /root/reference/lib/cmex2compute.m, which is what the reference's unit tests
drive (examples/testing/testLinearAlgebra.m:12, testFactorizations.m:74-90, testLDL.m:84-92).
Used here by the parity tests of the individual scalarizers."""
from __future__ import annotations

from .csparse import CSparse


def compute_tape(inputs, outputs, classname="tmpC"):
    """inputs: list of Tvariables; outputs: dict name->expression.  Returns the Tape with entry
    points ``<cls>_set_<var>`` and ``<cls>_get_<name>``."""
    code = CSparse()
    for v in inputs:
        code.declareSet(v, f"{classname}_set_{v.par}", external=True)
    for nm, e in outputs.items():
        code.declareGet([e], f"{classname}_get_{nm}", [nm])
    return code.compile({}), code
