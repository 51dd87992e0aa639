"""GPU: BASELINE configs[2..4] at their FULL batch sizes, through properties that do not need the oracle to solve tens of
thousands of instances (the oracle comparisons at small batch sizes are tests/test_golden_configs.py / test_gpu_interp.py):

* replication invariance -- the batch tiles 256 distinct seeded instances, so instance k and instance k+256 get the same
  inputs at different positions of the batch (another wavefront, another CU): status, iteration count and every output must
  be BITWISE equal;
* every instance converges (status 0);
* the returned objective equals the objective recomputed from the returned primal variables with a formula written by hand
  (tests/dense_ipm.py, not the tape), and the returned point satisfies the constraints to the solver's tolerances;
* the first instances equal the committed golden vectors where their seeds coincide (l1l2, dcmotor: bit for bit)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
NDISTINCT = 256


def _run(which, variant, B, seed0=0):
    import __graft_entry__ as g
    from tenscalc_amd.runtime import HipSolver
    if which == "l1l2":
        from tenscalc_amd.problems.l1l2 import SOLVE_ARGS, l1l2 as mk, l1l2_data as data
    elif which == "dcmotor":
        from tenscalc_amd.problems.mpc_dcmotor import SOLVE_ARGS, dcmotor_data as data, mpc_dcmotor as mk
    else:
        from tenscalc_amd.problems.mpc_quadcopter import SOLVE_ARGS, mpc_quadcopter as mk, quadcopter_data as data
    spec = mk(folder=g.GENERATED)
    so = spec.build(**g.VARIANT_KW[variant])
    names = [n for n, _ in spec.parameters + spec.variables]
    nd = min(B, NDISTINCT)
    ds = [data(seed0 + k) for k in range(nd)]
    obj = HipSolver(so, B=B)
    for n in names:
        obj.set(n, np.stack([np.asarray(ds[k % nd][n], dtype=float).reshape(-1, order="F") for k in range(B)]))
    st, it, _ = obj.solve(**SOLVE_ARGS)
    out = obj.getOutputs(as_dict=True)
    obj.close()
    return spec, ds, st, it, {k: v.reshape(B, -1) for k, v in out.items()}


def _replicas_equal(st, it, out, B):
    nd = min(B, NDISTINCT)
    for r in range(1, B // nd):
        sl = slice(r * nd, (r + 1) * nd)
        assert np.array_equal(st[sl], st[:nd]) and np.array_equal(it[sl], it[:nd])
        for k, v in out.items():
            assert np.array_equal(v[sl], v[:nd], equal_nan=True), k


def test_l1l2_full_batch_properties():
    from dense_ipm import l1l2_qp
    B = 16384
    spec, ds, st, it, out = _run("l1l2", "interp8", B, seed0=10)        # seeds 10..15 are the golden ones
    assert (st == 0).all() and it.max() <= 100
    _replicas_equal(st, it, out, B)
    for k in (0, 3, 100, 255):
        P, q, r, _, _, C, dd, _ = l1l2_qp(ds[k])
        u = np.r_[out["position"][k], out["noise1"][k], out["acceleration1"][k], out["noise1abs"][k], out["acceleration1abs"][k]]
        J = .5 * u @ (P @ u) + q @ u + r
        assert abs(out["J"][k, 0] - J) <= 1e-9 * (1 + abs(J))
        assert (C @ u + dd).min() > 0                                    # interior point: strictly feasible
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "l1l2.npz"))
    n = len(gold["seeds"])
    assert list(gold["seeds"]) == list(range(10, 10 + n))
    got = np.concatenate([out[nm][:n] for nm in gold["output_names"]], axis=1)
    assert np.array_equal(st[:n], gold["status"]) and np.array_equal(it[:n], gold["iter"])
    assert np.array_equal(got, gold["outputs"])


def test_dcmotor_full_batch_properties():
    from dense_ipm import dcmotor_qp
    B = 32768
    spec, ds, st, it, out = _run("dcmotor", "ldscs", B, seed0=100)       # bench.py's default; seeds 100..131 are the golden ones
    assert (st == 0).all() and it.max() <= 100
    _replicas_equal(st, it, out, B)
    for k in (0, 17, 255):
        P, q, r, E, e, C, dd, _ = dcmotor_qp(ds[k])
        u = np.r_[out["u"][k], out["x"][k]]
        J = .5 * u @ (P @ u) + q @ u + r
        assert abs(out["J"][k, 0] - J) <= 1e-9 * (1 + abs(J))
        assert np.abs(E @ u + e).max() <= 1e-4 and (C @ u + dd).min() > 0          # equalTolerance; interior point
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dcmotor.npz"))
    n = len(gold["seeds"])
    got = np.concatenate([out[nm][:n] for nm in gold["output_names"]], axis=1)
    assert np.array_equal(st[:n], gold["status"]) and np.array_equal(it[:n], gold["iter"])
    assert np.array_equal(got, gold["outputs"])
    # the per-GPU share of the 8-GPU sweep on the printed-code classes (one thread per instance; 32 lanes per instance):
    # bit for bit what the LDS machine returns
    B2 = 4096
    for other in ("strict", "w32x"):
        _, _, st2, it2, out2 = _run("dcmotor", other, B2, seed0=100)
        assert np.array_equal(st2, st[:B2]) and np.array_equal(it2, it[:B2])
        for k, v in out2.items():
            assert np.array_equal(v, out[k][:B2]), (other, k)


@pytest.mark.parametrize("B", [512, 4096])
def test_quadcopter_full_batch_properties(B):
    """B = 4 096 is BASELINE configs[4] as stated (on one GPU: the workgroup dispatcher hands 4 096 instances to 256 CUs as
    they finish), 512 the per-GPU share of its 8-GPU form.  At 4 096 the 256 distinct seeds appear 16 times each, at
    different positions of the dispatch order."""
    import torch
    from dense_ipm import quadcopter_model
    spec, ds, st, it, out = _run("quadcopter", "ldsc4w", B, seed0=0)       # bench.py's default for this config
    assert (st == 0).all() and it.max() <= 1000
    _replicas_equal(st, it, out, B)
    for k in (0, 41, 255):
        model, _ = quadcopter_model(ds[k])
        z = torch.as_tensor(np.r_[out["p"][k], out["u"][k], out["positive2"][k]])
        assert abs(out["J"][k, 0] - float(model["f"](z))) <= 1e-9 * (1 + abs(out["J"][k, 0]))
        assert float(model["G"](z).abs().max()) <= 1e-4 and float(model["F"](z).min()) > 0


def test_profiling_view_through_the_boundary(tmp_path):
    """the reference's profilingView(filename) (writeCprofiling.c:8-83) on a class built with stats=True (the reference's
    profiling=true): one row per dependency group with executions and time, totals, phases; ordinary classes refuse;
    the counters do not change any result"""
    import __graft_entry__ as g
    from tenscalc_amd.problems.sls import SOLVE_ARGS, sls_batch
    from tenscalc_amd.runtime import HipSolver
    spec = g.sls_spec("strict")
    B = 256
    A, b, x0 = sls_batch(B, seed0=500)
    res = {}
    for tag, kw in (("plain", g.VARIANT_KW["strict"]), ("stats", dict(g.VARIANT_KW["strict"], stats=True, tag="strict_stats"))):
        obj = HipSolver(spec.build(**kw), B=B)
        obj.set("A", np.ascontiguousarray(np.transpose(A, (0, 2, 1)))); obj.set("b", b); obj.set("x", x0)
        st, it, _ = obj.solve(**SOLVE_ARGS)
        res[tag] = (st, it, [o.copy() for o in obj.getOutputs()])
        path = str(tmp_path / (tag + ".profile"))
        if tag == "plain":
            with pytest.raises(RuntimeError, match="without profiling"):
                obj.profilingView(path)
        else:
            obj.profilingView(path)
            txt = open(path).read().splitlines()
            assert txt[0].split() == ["Operation", "Count"] and txt[18].split()[0] == "Total"      # writeCprofiling.c:16-22
            txt = txt[[k for k, ln in enumerate(txt) if ln.startswith("Group ")][0]:]
            rows = [ln.split() for ln in txt[1:1 + spec.tape.nGroups]]
            assert [int(r[0]) for r in rows] == list(range(spec.tape.nGroups))
            execs = np.array([float(r[2]) for r in rows]); tus = np.array([float(r[3]) for r in rows])
            assert execs.max() >= it.mean() - 2 and (tus >= 0).all() and tus.sum() > 0
            assert txt[1 + spec.tape.nGroups].startswith("Total") and any(ln.startswith("Phase of the Newton loop") for ln in txt)
        obj.close()
    assert np.array_equal(res["plain"][0], res["stats"][0]) and np.array_equal(res["plain"][1], res["stats"][1])
    for a, b_ in zip(res["plain"][2], res["stats"][2]):
        assert np.array_equal(a, b_)


def test_sls_large_ragged_batch_properties():
    """sls beyond the benchmark size and not a multiple of anything (B = 150 001, 1.1 GB of inputs): the head of the batch
    equals the golden vectors bit for bit, replicated instances (k, k + 64 000) are bitwise equal, all converge, the
    returned x is inside the box"""
    import __graft_entry__ as g
    from tenscalc_amd.problems.sls import SOLVE_ARGS, sls_batch
    from tenscalc_amd.runtime import HipSolver
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sls_N100_n8.npz"))
    n0, per, B = len(gold["seeds"]), 64000, 150001
    A, b, x0 = sls_batch(per, seed0=int(gold["seeds"][0]))
    idx = np.arange(B) % per
    obj = HipSolver(g.sls_spec(g.BENCH_DEFAULT).build(**g.VARIANT_KW[g.BENCH_DEFAULT]), B=B)
    obj.set("A", np.ascontiguousarray(np.transpose(A, (0, 2, 1)))[idx]); obj.set("b", b[idx]); obj.set("x", x0[idx])
    st, it, _ = obj.solve(**SOLVE_ARGS)
    J, x = (o.reshape(B, -1) for o in obj.getOutputs())
    obj.close()
    assert (st == 0).all()
    assert np.array_equal(st[:n0], gold["status"]) and np.array_equal(it[:n0], gold["iter"])
    assert np.array_equal(J[:n0, 0], gold["J"]) and np.array_equal(x[:n0], gold["x"])
    for r in (1, 2):
        k = min(per, B - r * per)
        assert np.array_equal(it[r * per:r * per + k], it[:k]) and np.array_equal(J[r * per:r * per + k], J[:k])
        assert np.array_equal(x[r * per:r * per + k], x[:k])
    assert x.min() > 0 and x.max() < .05
