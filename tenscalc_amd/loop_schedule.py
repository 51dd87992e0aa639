"""Which dependency groups are certainly clean at each entry-point call of the Newton loop.

The reference evaluates dependency groups lazily: every ``get``/``copy`` tests the status flag of
each group it needs (writeCfunctionpergroup.m:653-657) and evaluates the dirty ones.  The printed-code
back-ends inline the groups into the entry points and the entry points into the loop, so a group
needed by k entry points is k copies of its code in the kernel -- although the loop calls the entry
points in a fixed order and, e.g., the KKT factorisation evaluated for ``getCurvature__`` is always
still clean when ``getDirectionError__``, ``getMaxAlphas_s__``, ``getMinF_s__`` and
``updatePrimalDual__`` need it.

This module restates the call structure of csrc/ipm_newton.hip.inc (one call site per get, loops
and conditionals only -- no arithmetic) and runs a forward "must be clean" data-flow analysis over
it: a group is *certainly clean* at a call site if on every path reaching the site it has been
evaluated after the last set/copy that could invalidate it.  Such a group needs neither its test
nor its code at that site.  The flags stay in the program for everything else, so the analysis
only has to be sound, not complete; ``hip_emitter`` keeps unelided twins of all entry points for
the probe kernel and for getOutputs, which run outside the loop's context.

Soundness is checked end to end: the ``strict`` class is built WITH the elision and must remain
bit-identical to the oracle (which keeps the dynamic flags) on every GPU parity test.
"""
from __future__ import annotations

__all__ = ["newton_program", "certainly_clean", "analyse"]


def newton_program(d):
    """Call structure of ipmPD_CSsolver() in csrc/ipm_newton.hip.inc for the #define block ``d``.
    Nodes: ("call", name) | ("if", [nodes]) | ("loop", [nodes]) | ("exit",).  A loop that contains
    ("exit",) markers is left only there; a loop without markers may be left after any statement."""
    nF, nG = int(d["nF"]), int(d["nG"])
    call = lambda n: ("call", n)                                    # noqa: E731
    P = []
    if d["setAddEye2Hessian"]:
        P += [call("setAddEye2HessianU__"), call("setAddEye2HessianEq__")]
    P.append(call("initPrimal__"))
    if d["scaleCost"]:
        P += [call("scaleCost__"), call("getScale4Cost__")]
    if nF > 0:
        if d["scaleInequalities"]:
            P.append(call("scaleIneq__"))
        P += [call("setMu__"), call("initDualIneq__")]
    if nG > 0:
        P.append(call("initDualEqX__"))
    body = [call("getNorminf_Grad__")]
    if nG > 0:
        body.append(call("getNorminf_G__"))
    if nF > 0:
        body.append(call("getGapMinFMinLambda__"))
    if d["setAddEye2Hessian"] and d["adjustAddEye2Hessian"]:
        body += [("if", [call("setAddEye2HessianU__")]), ("if", [call("setAddEye2HessianEq__")])]
        inner = [call("getCurvature__")]
        if d["useInertia"]:
            inner.append(("if", [call("getHessInertia__")]))
        inner.append(call("getDirectionError__"))
        # left after pass 1, when the curvature test passes, or when no regularisation changed: in all
        # three cases before any set of this pass; after a set the loop continues with pass 1
        inner.append(("exit",))
        inner.append(("if", [call("setAddEye2HessianU__")]))
        inner.append(("if", [call("setAddEye2HessianEq__")]))
        body.append(("loop", inner))
    if nF == 0:
        body.append(call("setAlphaPrimal__"))
        if nG > 0:
            body.append(call("setAlphaDualEq__"))
        body.append(call("updatePrimalDual__"))
    else:
        if not int(d.get("skipAffine", 1)):       # predictor step (ipm_newton.hip.inc, reference :580-664)
            body.append(call("getMaxAlphas_a__"))
            body.append(("loop", [call("setAlphaPrimal__"), ("if", [call("getMinF_a__")])]))
            body.append(call("setAlphaDualIneq__"))
            body.append(("if", [call("getRho__"), call("setMu__")]))
        body.append(call("getMaxAlphas_s__"))
        body.append(("loop", [call("setAlphaPrimal__"), ("if", [call("getMinF_s__")])]))
        if nG > 0:
            body.append(call("setAlphaDualEq__"))
        body += [call("setAlphaDualIneq__"), call("updatePrimalDual__"),
                 ("if", [call("setMu__")]), ("if", [call("setMu__"), call("initDualIneq__")]),
                 ("if", [call("setMu__")])]
    P.append(("loop", body))
    return P


def certainly_clean(tape, defines):
    """{entry name: set of groups (1-based) that are clean at EVERY call of that entry in the loop}."""
    return analyse(tape, defines)[0]


def analyse(tape, defines, pre_clean=()):
    """(clean, dirty): per entry point, the groups that are certainly clean / certainly dirty at
    every one of its call sites.  A certainly dirty group is evaluated there without testing its
    flag -- which also tells the compiler that last iteration's values of that group are dead."""
    ent = {}
    for e in tape.sets:
        ent[e["name"]] = (frozenset(), frozenset(e["children"]))
    for e in tape.gets:
        ent[e["name"]] = (frozenset(e["parents"]), frozenset())
    for e in tape.copies:
        ent[e["name"]] = (frozenset(e["parents"]), frozenset(e["children"]))
    sites_c, sites_d = {}, {}
    heads = []           # fixpoint state at the head of every loop, outermost last
    backs = []           # ... and at the end of its body (the back edge)

    def meet(a, b):
        return (a[0] & b[0], a[1] & b[1])

    def run(nodes, st, record):
        clean, dirty = st
        for nd in nodes:
            if nd[0] == "call":
                parents, children = ent.get(nd[1], (frozenset(), frozenset()))
                if parents:
                    if record:
                        sc, sd = clean & parents, dirty & parents
                        sites_c[nd[1]] = sc if nd[1] not in sites_c else sites_c[nd[1]] & sc
                        sites_d[nd[1]] = sd if nd[1] not in sites_d else sites_d[nd[1]] & sd
                    clean, dirty = clean | parents, dirty - parents
                if children:
                    clean, dirty = clean - children, dirty | children
            elif nd[0] == "if":
                clean, dirty = meet((clean, dirty), run(nd[1], (clean, dirty), record))
            elif nd[0] == "loop":  # any number of passes
                head = (clean, dirty)
                while True:
                    new = meet(head, run(nd[1], head, False))
                    if new == head:
                        break
                    head = new
                back = run(nd[1], head, record)
                if record:
                    heads.append(head)
                    backs.append(back)
                marks = [k for k, x in enumerate(nd[1]) if x[0] == "exit"]
                if marks:      # left only at the markers (at least one pass)
                    sts = [run(nd[1][:k], head, False) for k in marks]
                    clean, dirty = sts[0]
                    for x in sts[1:]:
                        clean, dirty = meet((clean, dirty), x)
                else:          # what holds at EVERY point of the body (break anywhere, zero passes too)
                    clean, dirty = meet(head, _everywhere(nd[1], head, run, meet))
        return (clean, dirty)

    allg = frozenset(range(1, tape.nGroups + 1))
    pre = frozenset(pre_clean)       # groups the kernel prologue has already evaluated (parameter-only ones)
    run(newton_program(defines), (pre, allg - pre), True)
    analyse.loop_head_dirty = set(heads[-1][1]) if heads else set()   # at the head of the Newton loop itself
    analyse.loop_back_dirty = set(backs[-1][1]) if backs else set()   # at the end of a pass: nothing of these
    #                                                                   groups is carried into the next pass
    return {k: set(v) for k, v in sites_c.items()}, {k: set(v) for k, v in sites_d.items()}


def _everywhere(nodes, st, run, meet):
    acc = st
    for k in range(len(nodes)):
        acc = meet(acc, run(nodes[:k + 1], st, False))
        if nodes[k][0] in ("if", "loop"):
            acc = meet(acc, _everywhere(nodes[k][1], run(nodes[:k], st, False), run, meet))
    return acc


# ---------------------------------------------------------------------------------------------------
# life times of the values of transient groups (for slot sharing, lowering.Lowering(share=...))
# ---------------------------------------------------------------------------------------------------
def timeline(tape, defines):
    """Linear time over one pass of the Newton loop body.

    Returns dict(time={entry name: [positions]}, loops=[(first, last)] of the inner loops,
    T_init=-1, T_end, E={group: sorted positions where it may be evaluated}, transient=set of groups
    certainly dirty at the loop head).  Position -1 stands for the initialisation sequence, T_end for
    what runs after the loop (getOutputs).  transient = groups certainly dirty at the END of a pass:
    none of their values is carried into the next pass (the first pass may still find them clean from the
    initialisation sequence, position -1)."""
    clean, dirty = analyse(tape, defines)
    transient = set(analyse.loop_back_dirty)
    prog = newton_program(defines)
    body = [nd for nd in prog if nd[0] == "loop"][-1][1]
    init = [nd for nd in prog if nd[0] == "call"]
    time, loops = {}, []
    pos = [0]

    def walk(nodes):
        for nd in nodes:
            if nd[0] == "call":
                time.setdefault(nd[1], []).append(pos[0])
                pos[0] += 1
            elif nd[0] == "if":
                walk(nd[1])
            elif nd[0] == "loop":
                a = pos[0]
                walk(nd[1])
                loops.append((a, pos[0] - 1))
    walk(body)
    T_end = pos[0]
    for nd in init:
        time.setdefault(nd[1], []).append(-1)
    ent = {e["name"]: e for e in list(tape.gets) + list(tape.copies)}
    E = {}
    for nm, ts in time.items():
        e = ent.get(nm)
        if e is None:
            continue
        for g in e["parents"]:
            if g not in clean.get(nm, ()):
                E.setdefault(g, set()).update(ts)
    # also transient: a group that is re-evaluated before every use -- certainly dirty at the first call of a
    # pass that needs it, and elsewhere certainly clean only thanks to that evaluation earlier in the SAME
    # pass; such a group never relies on what its storage held before (e.g. mu./F, evaluated for
    # initDualIneq__ right after setMu__ and read by nothing else)
    uses = {}
    for nm, ts in time.items():
        e = ent.get(nm)
        if e is None:
            continue
        for g in e["parents"]:
            what = "dirty" if g in dirty.get(nm, ()) else ("clean" if g in clean.get(nm, ()) else "dynamic")
            for t in ts:
                uses.setdefault(g, []).append((t, what))
    for g, us in uses.items():
        if g in transient or any(w == "dynamic" for _, w in us):
            continue
        first_dirty = min([t for t, w in us if w == "dirty" and t >= 0], default=None)
        ok = all(w == "dirty" or (t >= 0 and first_dirty is not None and t > first_dirty) for t, w in us)
        if ok:
            transient.add(g)
    return dict(time=time, loops=loops, T_init=-1, T_end=T_end, E={g: sorted(v) for g, v in E.items()},
                transient=transient, clean=clean, dirty=dirty)
