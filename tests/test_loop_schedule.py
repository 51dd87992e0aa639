"""CPU: soundness of tenscalc_amd/loop_schedule.py.  The DEVICE Newton loop compiled for the host
(oracle/devloop_glue.cpp) records the entry points it calls; replaying that sequence through the
reference's lazy-evaluation protocol (flag per group: a get/copy evaluates its dirty parents, a
set/copy invalidates its children, writeCfunctionpergroup.m:513-852) must never contradict a static
claim: a group claimed certainly clean at an entry point is clean at every call of it, a group claimed
certainly dirty is dirty.  Also: the analysis actually elides something (it is not vacuous)."""
import numpy as np
import pytest

from tenscalc_amd.loop_schedule import analyse
from tenscalc_amd.problems.sls import sls_data
from test_solver_options import OPTIONS, _problem


def _replay(tape, names, clean, dirty):
    ent = {}
    for e in tape.sets:
        ent[e["name"]] = ((), tuple(e["children"]))
    for e in tape.gets:
        ent[e["name"]] = (tuple(e["parents"]), ())
    for e in tape.copies:
        ent[e["name"]] = (tuple(e["parents"]), tuple(e["children"]))
    is_dirty = {g: True for g in range(1, tape.nGroups + 1)}     # kernel start: every flag cleared
    checked = 0
    for nm in names:
        parents, children = ent[nm]
        for p in parents:
            if p in clean.get(nm, ()):
                assert not is_dirty[p], (nm, p, "claimed clean, is dirty")
                checked += 1
            if p in dirty.get(nm, ()):
                assert is_dirty[p], (nm, p, "claimed dirty, is clean")
                checked += 1
        for p in parents:
            is_dirty[p] = False
        for c in children:
            is_dirty[c] = True
    return checked


def test_sls_traces_never_contradict_the_static_claims(sls_spec, sls_tape_path):
    from oracle.tc_oracle import DeviceLoopOracle
    do = DeviceLoopOracle(sls_tape_path, "sls", sls_spec.defines)
    clean, dirty = analyse(sls_spec.tape, sls_spec.defines)
    assert sum(len(v) for v in clean.values()) > 50 and sum(len(v) for v in dirty.values()) > 5
    c = sls_spec.classname
    total = 0
    for seed, (mu0, maxIter, eps) in enumerate([(1.0, 30, 1e-9), (1e-3, 30, 1e-9), (1.0, 3, 1e-9), (10.0, 60, 1e-3),
                                                (1.0, 30, 1e-20), (1e3, 12, 1e-9)] * 4):
        A, b, x0 = sls_data(500 + seed)
        if seed % 7 == 3:
            A = A * 1e6                     # regularisation retries
        if seed % 11 == 5:
            x0 = np.full(8, .06)            # infeasible start: immediate exit
        do.set(c + "_set_A", A); do.set(c + "_set_b", b); do.set(c + "_set_x", x0)
        do.solve(mu0, maxIter, -1, (eps, eps))
        total += _replay(sls_spec.tape, do.trace(), clean, dirty)
    assert total > 10000


@pytest.mark.parametrize("kind", ["box", "eq", "eqonly", "free"])
@pytest.mark.parametrize("opt", OPTIONS, ids=[str(k) for k in range(len(OPTIONS))])
def test_template_branches(native_built, tmp_path, kind, opt):
    from tenscalc_amd.tapeio import save_tape
    from oracle.tc_oracle import DeviceLoopOracle
    if kind in ("eqonly", "free") and ("scaleInequalities" in opt or "coupledAlphas" in opt):
        pytest.skip("option only matters with inequality constraints")
    spec = _problem(kind, **opt)
    tape = str(tmp_path / "p.tape")
    save_tape(spec.tape, tape)
    do = DeviceLoopOracle(tape, "sched_%s_%d" % (kind, OPTIONS.index(opt)), spec.defines)
    clean, dirty = analyse(spec.tape, spec.defines)
    c = spec.classname
    rs = np.random.RandomState(4)
    total = 0
    for trial in range(8):
        A, b, x0 = rs.random_sample((12, 4)), rs.random_sample(12), .02 * rs.random_sample(4) + .005
        if trial >= 6:
            A = A * 1e4
        do.set(c + "_set_A", A); do.set(c + "_set_b", b); do.set(c + "_set_x", x0)
        do.solve(1.0 if trial % 2 else 1e-2, 40 if trial != 5 else 3, -1, (1e-9, 1e-9))
        total += _replay(spec.tape, do.trace(), clean, dirty)
    assert total > 0
