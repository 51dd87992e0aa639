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
    # addEye2Hessian < 0 makes the curvature negative: the only way this convex problem reaches the second pass
    # of the regularisation loop (sets + re-evaluation inside one Newton iteration)
    for seed, (mu0, maxIter, eps) in enumerate([(1.0, 30, 1e-9), (1e-3, 30, 1e-9), (1.0, 3, 1e-9), (10.0, 60, 1e-3),
                                                (1.0, 30, 1e-20), (1e3, 12, 1e-9), (1.0, 30, -10.0), (1.0, 30, -1.0)] * 3):
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
    for trial in range(10):
        A, b, x0 = rs.random_sample((12, 4)), rs.random_sample(12), .02 * rs.random_sample(4) + .005
        if trial in (6, 7):
            A = A * 1e4
        ae = (-5.0, -5.0) if trial >= 8 else (1e-9, 1e-9)     # negative: second regularisation pass
        do.set(c + "_set_A", A); do.set(c + "_set_b", b); do.set(c + "_set_x", x0)
        do.solve(1.0 if trial % 2 else 1e-2, 40 if trial != 5 else 3, -1, ae)
        total += _replay(spec.tape, do.trace(), clean, dirty)
    assert total > 0


# ---------------------------------------------------------------------------------------------------
# slot sharing (Lowering._assign_shared_slots) replayed on real call traces
# ---------------------------------------------------------------------------------------------------
def _replay_storage(tape, L, names):
    """Walk a recorded entry-point sequence through the lazy-evaluation protocol with an explicit model of
    the shared storage: owner[slot] = the instruction whose value the slot holds.  Every operand an
    evaluated instruction reads from storage, and every value an entry point reads itself, must still be
    owned by the instruction that is supposed to be there."""
    ent = {}
    for e in tape.sets:
        ent[e["name"]] = ("set", e)
    for e in tape.gets:
        ent[e["name"]] = ("get", e)
    for e in tape.copies:
        ent[e["name"]] = ("copy", e)
    is_dirty = {g: True for g in range(1, tape.nGroups + 1)}
    owner = {}
    checked = 0

    def read(j, who):
        nonlocal checked
        k = int(L.pslot[j])
        if k >= 0:
            assert owner.get(k) == j, (who, "reads instruction", j, "from slot", k, "which holds", owner.get(k))
            checked += 1

    def evaluate(g):
        for i in tape.groups[g - 1].tolist():
            if not tape.used[i]:
                continue
            for j in L.operands(i):
                j = int(j)
                if tape.isload[j - 1] or L.external[j]:
                    continue
                if L.local[j] or L.inline[j]:
                    continue                      # lives in a register inside its group
                read(j, ("group", g, "instruction", i))
            if L.pslot[i] >= 0:
                owner[int(L.pslot[i])] = i

    for nm in names:
        kind, e = ent[nm]
        if kind in ("get", "copy"):
            for g in e["parents"]:
                if is_dirty[g]:
                    evaluate(g)
                    is_dirty[g] = False
        if kind == "get":
            for o in e["outputs"]:
                for j in o["ins"]:
                    read(int(j), nm)
        elif kind == "copy":
            for d, s_ in zip(e["dst"], e["src"]):
                read(int(s_), nm)
            for d in e["dst"]:
                if L.pslot[d] >= 0:
                    owner[int(L.pslot[d])] = int(d)
        else:
            for i in e["ins"]:
                if L.pslot[i] >= 0:
                    owner[int(L.pslot[i])] = int(i)
        for c in e.get("children", []):
            is_dirty[c] = True
    return checked


@pytest.mark.parametrize("fine", [False, True], ids=["coarse", "fine"])
@pytest.mark.parametrize("lanes", [1, 8])
def test_shared_slots_never_hold_the_wrong_value_on_sls_traces(sls_spec, sls_tape_path, lanes, fine, monkeypatch):
    from tenscalc_amd.lowering import Lowering
    from oracle.tc_oracle import DeviceLoopOracle
    monkeypatch.setattr(Lowering, "FINE_LIVES", fine)      # both life-time models (the finer one is opt-in)
    L = Lowering(sls_spec.tape, sls_spec.code, roll=True, lanes=lanes, share=sls_spec.defines)
    L0 = Lowering(sls_spec.tape, sls_spec.code, roll=True, lanes=lanes)
    assert L.nPriv < 0.9 * L0.nPriv                                   # sharing actually happens
    do = DeviceLoopOracle(sls_tape_path, "sls", sls_spec.defines)
    c = sls_spec.classname
    total = 0
    second = 0
    for seed, (mu0, maxIter, eps) in enumerate([(1.0, 30, 1e-9), (1e-3, 30, 1e-9), (1.0, 3, 1e-9), (10.0, 60, 1e-3),
                                                (1.0, 30, -10.0), (1.0, 30, -1.0)] * 3):
        A, b, x0 = sls_data(900 + seed)
        if seed % 5 == 3:
            A = A * 1e6
        do.set(c + "_set_A", A); do.set(c + "_set_b", b); do.set(c + "_set_x", x0)
        do.solve(mu0, maxIter, -1, (eps, eps))
        # the kernel: external sets, the loop, then getOutputs
        tr = do.trace()
        second += sum(1 for a, b_ in zip(tr, tr[1:]) if a.startswith("setAddEye2Hessian") and b_ == "getCurvature__")
        names = [c + "_set_A", c + "_set_b", c + "_set_x"] + tr + [c + "_getOutputs"]
        total += _replay_storage(sls_spec.tape, L, names)
    assert total > 50000 and second > 0          # second regularisation passes were exercised


@pytest.mark.parametrize("kind", ["box", "eq", "eqonly", "free"])
@pytest.mark.parametrize("opt", OPTIONS[:4], ids=[str(k) for k in range(4)])
def test_shared_slots_on_template_branches(native_built, tmp_path, kind, opt):
    from tenscalc_amd.lowering import Lowering
    from tenscalc_amd.tapeio import save_tape
    from oracle.tc_oracle import DeviceLoopOracle
    if kind in ("eqonly", "free") and ("scaleInequalities" in opt or "coupledAlphas" in opt):
        pytest.skip("option only matters with inequality constraints")
    spec = _problem(kind, **opt)
    tape = str(tmp_path / "p.tape")
    save_tape(spec.tape, tape)
    do = DeviceLoopOracle(tape, "shr_%s_%d" % (kind, OPTIONS.index(opt)), spec.defines)
    Ls = []
    for fine in (False, True):
        Lowering.FINE_LIVES = fine
        Ls.append(Lowering(spec.tape, spec.code, roll=True, lanes=1, share=spec.defines))
    Lowering.FINE_LIVES = False
    c = spec.classname
    rs = np.random.RandomState(5)
    for trial in range(8):
        A, b, x0 = rs.random_sample((12, 4)), rs.random_sample(12), .02 * rs.random_sample(4) + .005
        if trial in (4, 5):
            A = A * 1e4
        ae = (-5.0, -5.0) if trial >= 6 else (1e-9, 1e-9)     # negative: second regularisation pass
        do.set(c + "_set_A", A); do.set(c + "_set_b", b); do.set(c + "_set_x", x0)
        do.solve(1.0 if trial % 2 else 1e-2, 40 if trial != 3 else 3, -1, ae)
        names = [c + "_set_A", c + "_set_b", c + "_set_x"] + do.trace() + [c + "_getOutputs"]
        for L in Ls:
            assert _replay_storage(spec.tape, L, names) > 0
