#!/usr/bin/env python3
"""Headline benchmark: IPM solves/sec, sls N=100 n=8, batch 65536 per GPU (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (whole-batch interior-point solve, one kernel launch) over one
batch of synthetic problem instances already resident in HBM in structure-of-arrays layout.  The
batch shards embarrassingly: every rank owns B instances on its own GPU (weak scaling) and the only
collective is the gather of results (outputs + status) over RCCL, included in the timed region for
N>1.  Rank 0 prints ONE JSON line (see DESIGN.md "Measurement" for the roofline arithmetic).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_SOLVE = 8 * (800 + 100 + 8) + 8 * (1 + 8) + 8   # 7344 B
HBM_PEAK_GBPS = 8000.0                                           # MI355X_MICROARCH.md chip table


class _DevArray:
    """zero-copy view of handle-owned device memory for torch (cuda array interface v2)"""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = dict(shape=tuple(shape), typestr=typestr, data=(int(ptr), False),
                                             version=2, strides=None)


def make_batch(B, rank, dev):
    """the benchmark batch of rank `rank`: A [800][B] (column-major A of each instance), b [100][B], x0 [8][B],
    structure-of-arrays on `dev` (the distributions of examples/sls.m:36-41); shared with
    tests/test_gpu_parity.py, which checks the default variant bit for bit on instances of THIS batch"""
    import torch
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    A = torch.rand((800, B), generator=gen, device=dev, dtype=torch.float64)
    b = torch.rand((100, B), generator=gen, device=dev, dtype=torch.float64)
    x0 = .02 * torch.rand((8, B), generator=gen, device=dev, dtype=torch.float64)
    return A, b, x0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=0, help="instances per GPU (default: the config's)")
    ap.add_argument("--variant", default=os.environ.get("TC_BENCH_VARIANT", ""),
                    help="w8x (default): 8 lanes/instance wave-cooperative lowering of the REFERENCE tape with the "
                         "reference's arithmetic kept bit for bit (parity class bit-exact: status, iteration count and "
                         "outputs identical to the cmex reference on every instance); strict: 1 thread/instance, "
                         "-ffp-contract=off, bit-exact too; fast: same with FMA; w8: the wave-cooperative lowering with "
                         "FMA, reassociated reductions and fast division; w8s: the same of the hoisted tape (parameter-only "
                         "contractions first, tenscalc_amd/hoist.py), inputs streamed, LDS slots shared, 6 wavefronts per "
                         "CU -- parity class solver-tolerance; w8h / regs / regsl: other lowerings of the hoisted tape")
    ap.add_argument("--cpu-sample", type=int, default=int(os.environ.get("TC_CPU_SAMPLE", "65536")))
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--config", default="sls", choices=["sls", "l1l2", "dcmotor", "quadcopter"],
                    help="sls (default): BASELINE.json's metric, configs[1].  The others are BASELINE configs[2..4] in the "
                         "same JSON vocabulary: l1l2 16384 instances, dcmotor 32768 (the whole 8-GPU sweep on one GPU), "
                         "quadcopter 512 (configs[4]'s per-GPU share of 4096) per GPU; --batch overrides")
    ap.add_argument("--pipeline", type=int, default=1,
                    help="--config runs only: number of batches in flight (that many solver handles, each on a stream of "
                         "its own, steps issued round-robin).  1 (default) = one batched solve after the other.  With >1 the "
                         "slowest instance of a batch -- mpc_quadcopter: 371 Newton iterations against a median of 28 -- no "
                         "longer idles the other 255 CUs: they start on the next batch.  Every step is still one complete "
                         "batched solve; reported under config.pipeline")
    ap.add_argument("--gather-every-step", action="store_true",
                    help="N>1: all-gather the results after EVERY step (overlapped with the next solve) instead of once "
                         "after the K steps (SURVEY.md 8e: one gather at the end)")
    args = ap.parse_args()
    if args.config != "sls":
        return bench_config(args)
    args.variant = args.variant or "w8x"
    import __graft_entry__ as g
    sls_variants = [v for v, kw in g.VARIANT_KW.items() if kw.get("body", "code") == "code" and v not in g.CONFIG_ONLY]
    if args.variant not in sls_variants:
        ap.error("--variant for --config sls: " + ", ".join(sls_variants))

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    single = bool(os.environ.get("TC_BENCH_SINGLE_DEVICE"))
    if single:   # rehearsal aid: several ranks on ONE GPU. RCCL refuses two ranks on a device, so the
        local = 0   # collectives go through gloo on host copies; never used for a reported number
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if single:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)   # RCCL on ROCm; one rank per GPU
    bar = (lambda: dist.barrier()) if single else (lambda: dist.barrier(device_ids=[local]))
    cdev = torch.device("cpu") if single else dev

    import __graft_entry__ as g
    from tenscalc_amd.problems.sls import SOLVE_ARGS
    from tenscalc_amd.runtime import LAYOUT_BATCH_MINOR, HipSolver
    spec = g.sls_spec(args.variant)
    kw = dict(g.VARIANT_KW[args.variant])
    if os.environ.get("TC_EXPERIMENT"):          # e.g. TC_EXPERIMENT="min_waves=2,block=256" (tuning runs)
        for kv in os.environ["TC_EXPERIMENT"].split(","):
            k_, v_ = kv.split("=")
            kw[k_] = int(v_)
        kw["tag"] = (kw.get("tag", "fast") + "_" + os.environ["TC_EXPERIMENT"].replace("=", "").replace(",", "_"))
    if rank == 0:
        so = spec.build(**kw)          # prebuilt in-tree normally; compiles here only if missing
    if world > 1:
        bar()
    so = spec.build(**kw)

    B = args.batch or 65536
    A, b, x0 = make_batch(B, rank, dev)
    obj = HipSolver(so, B=B, device=local)
    stream = torch.cuda.current_stream().cuda_stream
    for nm, t in (("A", A), ("b", b), ("x", x0)):
        obj.set_device(nm, t, LAYOUT_BATCH_MINOR, stream=stream)
    torch.cuda.synchronize()
    optr, sptr, iptr = obj.device_views()
    O = torch.as_tensor(_DevArray(optr, (9, B), "<f8"), device=dev)
    ST = torch.as_tensor(_DevArray(sptr, (B,), "<i4"), device=dev)
    IT = torch.as_tensor(_DevArray(iptr, (B,), "<i4"), device=dev)
    from tenscalc_amd.sharding import ResultExchange, pack_results, reduce_timing, run_timed
    # the path's only exchange: results (J, x, status, iter) of every rank gathered on every rank -- ONE gather after
    # the K steps by default; with --gather-every-step after every step, overlapped with the next
    # step's solve (packed on the launch stream, the collective asynchronous on RCCL's own stream).  The control
    # flow is tenscalc_amd/sharding.py:run_timed, rehearsed with gloo on the CPU by tests/test_sharding_gloo.py.
    every = world > 1 and args.gather_every_step
    xch = ResultExchange(world, (11, B), cdev) if world > 1 else None

    def solve(_k=0):
        obj.solve_async(SOLVE_ARGS["mu0"], SOLVE_ARGS["maxIter"], SOLVE_ARGS["addEye2Hessian"], stream=stream)

    for _ in range(args.warmup):
        solve()
        if every:
            xch.exchange(pack_results(O, ST, IT))
    if world > 1:
        if not every:      # one untimed gather: RCCL sets up its channels at the first collective of a kind and size
            xch.exchange(pack_results(O, ST, IT))
        xch.drain()
    # --- timed region: barrier + synchronize on both sides, exactly K steps
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]

    def step(k):
        evs[k][0].record()
        solve()
        evs[k][1].record()
    elapsed = run_timed(args.steps, step, lambda: pack_results(O, ST, IT), world, every=every,
                        barrier=bar, sync=torch.cuda.synchronize, exchange=xch)
    kern_ms = [a.elapsed_time(b_) for a, b_ in evs]          # events on the launch stream
    elapsed, solved_per_step = reduce_timing(elapsed, int((ST == 0).sum().item()), world, cdev)   # only status==0 count
    value = solved_per_step * args.steps / elapsed

    out = None
    if rank == 0:
        it = IT.cpu().numpy()
        kavg = float(np.mean(kern_ms)) * 1e-3
        achieved = ALGO_BYTES_PER_SOLVE * B / kavg / 1e9
        out = {
            "metric": "IPM solves/sec (whole node), sls N=100 n=8 batch",
            "value": value, "unit": "solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "sls box-constrained LS N=100 n=8, batch=%d random (A,b) per GPU, fp64 "
                                   "(BASELINE.json configs[1])" % B,
                       "batch_per_gpu": B, "mu0": SOLVE_ARGS["mu0"], "maxIter": SOLVE_ARGS["maxIter"],
                       "variant": args.variant,
                       "parity_class": g.PARITY_CLASS.get(args.variant, "solver-tolerance"),
                       "tape": ("hoisted (parameter-only contractions first); results within the stated tolerance of the "
                                "reference tape, tests/test_gpu_parity.py" if args.variant in g.HOISTED else "reference"),
                       "solved_fraction": solved_per_step / (B * world),
                       "iter_median": float(np.median(it)), "iter_max": int(it.max()),
                       "sharding": "independent instances per rank; RCCL all_gather of the results only, "
                                   + ("after every step, overlapped with the next solve" if every else "once after the K steps")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": measured_traffic(args.variant, B),
                         "kernel": "tc_solve_kernel", "kernel_ms_avg": kavg * 1e3,
                         "algorithmic_bytes_per_launch": ALGO_BYTES_PER_SOLVE * B,
                         "algorithmic_bytes_per_solve": ALGO_BYTES_PER_SOLVE,
                         # this path is sparse scalar fp64 work, so the vector-fp64 fraction is
                         # reported beside the HBM fraction the north star asks for
                         "fp64_valu": tape_flops(spec, it, B, kavg)},
        }
        if not args.no_cpu and world == 1:      # CPU baseline: rank 0 at N=1 only
            # always the reference's own tape (never the hoisted one): that is the path the reference runs
            out["cpu_baseline"] = cpu_baseline(g.sls_spec("strict"), A, b, x0, args.cpu_sample, SOLVE_ARGS)
        print(json.dumps(out), flush=True)
    if world > 1:
        bar()
        dist.destroy_process_group()
    obj.close()


FP64_VALU_PEAK_TFLOPS = 78.6       # MI355X_MICROARCH.md: vector fp64 (no MFMA on this path)


def tape_flops(spec, it, B, kernel_s):
    """Algorithmic flops of one launch from the tape's static per-group flop classes (the accounting of
    examples/flops.m:97-99 / instructionsTableUTHash.c:816-832): groups that depend on parameters only are
    evaluated once per solve, every other group once per Newton iteration, the groups behind
    getMinF_s__ once more per extra line-search trial (2 trials per iteration on this workload, measured with
    the stats build).  An estimate of useful work, not a counter."""
    T = spec.tape
    loop_sets = {"updatePrimalDual__", "setMu__", "setAlphaPrimal__", "setAlphaDualIneq__", "setAlphaDualEq__",
                 "setAddEye2HessianU__", "setAddEye2HessianEq__", "initDualIneq__", "initPrimal__"}
    varying = set()
    for e in list(T.sets) + list(T.copies):
        if e["name"] in loop_sets:
            varying |= set(e["children"])
    used = set()
    for e in list(T.gets) + list(T.copies):
        if e["name"] not in ("getGrad__", "getHessInertia__", "getJ__"):
            used |= set(e["parents"])
    once = sum(int(T.flops[g].sum()) for g in used if g not in varying)
    per_iter = sum(int(T.flops[g].sum()) for g in used if g in varying)
    minf = [e for e in T.gets if e["name"] == "getMinF_s__"]
    alpha = [e for e in T.sets if e["name"] == "setAlphaPrimal__"]
    trial = sum(int(T.flops[g].sum()) for g in (set(minf[0]["parents"]) & set(alpha[0]["children"]))) if minf and alpha else 0
    iters = float(it.mean())
    per_solve = once + iters * (per_iter + trial)
    tf = per_solve * B / kernel_s / 1e12
    return {"flops_per_solve": per_solve, "achieved": tf, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": tf / FP64_VALU_PEAK_TFLOPS}


def measured_traffic(variant, B):
    """HBM bytes per launch of tc_solve_kernel from the committed rocprofv3 PMC passes
    (profiles/traffic.json, written by scripts/profile_round.sh: separate --pmc FETCH_SIZE and
    --pmc WRITE_SIZE runs, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).
    None when no counter run exists for this exact variant and batch."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            t = json.load(f)
        e = t.get("%s_B%d" % (variant, B))
        return e["hbm_bytes_per_launch"] if e else None
    except (OSError, ValueError, KeyError):
        return None


def _cpu_worker(rank, n, q):
    """one process of the all-cores CPU baseline: its own copy of the generated-C twin (the reference's generated C
    keeps its state in globals: one instance stream per process, SURVEY.md 8b "Threading")"""
    import numpy as np
    import __graft_entry__ as g
    from oracle.c_emitter import CpuTwin
    from tenscalc_amd.problems.sls import SOLVE_ARGS
    tw = CpuTwin(g.sls_spec("strict"))
    rs = np.random.RandomState(100 + rank)
    A = rs.random_sample((n, 800)); b = rs.random_sample((n, 100)); x0 = .02 * rs.random_sample((n, 8))
    tw.solve_batch({"A": A[:64], "b": b[:64], "x": x0[:64]}, 64, mu0=SOLVE_ARGS["mu0"], maxIter=SOLVE_ARGS["maxIter"])
    t0 = time.perf_counter()
    out, st, it = tw.solve_batch({"A": A, "b": b, "x": x0}, n, mu0=SOLVE_ARGS["mu0"], maxIter=SOLVE_ARGS["maxIter"])
    q.put((rank, time.perf_counter() - t0, int((st == 0).sum())))


def cpu_all_cores(per_worker=30000):
    """the same CPU path on every host core this process may use (one worker process per core, each solving its
    own `per_worker` instances of the same distribution): aggregate solves/s, or None when the twin is not built"""
    import multiprocessing as mp
    P = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    P = max(1, min(P, int(os.environ.get("TC_CPU_WORKERS", "16"))))      # a one-GPU box's share of the host is 16 cores
    ps, res = [], []
    try:
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        ps = [ctx.Process(target=_cpu_worker, args=(r, per_worker, q)) for r in range(P)]
        for p_ in ps:
            p_.start()
        deadline = time.monotonic() + 300
        while len(res) < P:
            try:
                res.append(q.get(timeout=1.0))
            except Exception:          # queue.Empty: nothing yet -- is a worker gone (no prebuilt twin, killed)?
                if any(p_.exitcode not in (None, 0) for p_ in ps) or time.monotonic() > deadline:
                    return None
    except Exception:              # no all-cores figure
        return None
    finally:
        for p_ in ps:
            if p_.is_alive() and len(res) < P:
                p_.terminate()
        for p_ in ps:
            p_.join(timeout=10)
    wall = max(r[1] for r in res)
    return {"value_all_cores": sum(r[2] for r in res) / wall, "cores_all": P,
            "sample_all_cores": "%d worker processes x %d instances of the same distribution, slowest worker %.2f s"
                                % (P, per_worker, wall)}


def cpu_baseline(spec, A, b, x0, sample, solve_args):
    """The CPU restatement of the reference path timed on ONE host core over the first `sample`
    instances of the same workload (kind "port"), and on all host cores this process may use (one worker process
    per core).  Preferred: the generated-C twin (oracle/c_emitter.py: the tape as reference-style straight-line C +
    the reference's ipmPD_CSsolver.c, gcc -O1, prebuilt in the development container); fallback: the tape
    interpreter oracle/tc_oracle.c."""
    import numpy as np
    import __graft_entry__ as g
    c = spec.classname
    n = min(sample, A.shape[1])
    Ah = np.ascontiguousarray(A[:, :n].t().cpu().numpy())
    bh = np.ascontiguousarray(b[:, :n].t().cpu().numpy())
    xh = np.ascontiguousarray(x0[:, :n].t().cpu().numpy())
    allc = None
    try:
        from oracle.c_emitter import CpuTwin
        tw = CpuTwin(spec)
        tw.solve_batch({"A": Ah[:64], "b": bh[:64], "x": xh[:64]}, 64, mu0=solve_args["mu0"], maxIter=solve_args["maxIter"])
        t0 = time.perf_counter()
        reps = 0
        while reps == 0 or (time.perf_counter() - t0 < 10.0 and reps < 64):     # ~10 s of CPU work
            out, st, it = tw.solve_batch({"A": Ah, "b": bh, "x": xh}, n, mu0=solve_args["mu0"], maxIter=solve_args["maxIter"])
            reps += 1
        dt = (time.perf_counter() - t0) / reps
        what = ("generated-C twin of the tape (oracle/c_emitter.py) + reference lib/ipmPD_CSsolver.c, gcc -O1, "
                "prebuilt oracle/_ref/lib%s_cpu.so" % c)
        allc = cpu_all_cores()
    except (FileNotFoundError, OSError):
        from oracle.tc_oracle import Oracle
        o = Oracle(os.path.join(g.GENERATED, "@" + c, c + ".tape"))
        sets = lambda k: {c + "_set_A": Ah[:k], c + "_set_b": bh[:k], c + "_set_x": xh[:k]}  # noqa: E731
        o.solve_batch(sets(64), c + "_getOutputs", 64, mu0=solve_args["mu0"], maxIter=solve_args["maxIter"])
        t0 = time.perf_counter()
        out, st, it = o.solve_batch(sets(n), c + "_getOutputs", n, mu0=solve_args["mu0"], maxIter=solve_args["maxIter"])
        dt = time.perf_counter() - t0
        what = "oracle/tc_oracle.c (tape interpreter + restated ipmPD_CSsolver loop, gcc -O1)"
    res = {"value": float((st == 0).sum()) / dt, "unit": "solves/s", "cores": 1, "kind": "port",
           "sample": "first %d instances of the benchmark batch, %s, sequential on 1 of %d host cores, %.2f s per pass"
                     % (n, what, os.cpu_count(), dt),
           "us_per_solve": dt / n * 1e6}
    if allc:
        res.update(allc)
    return res


# ---------------------------------------------------------------------------------------------------------------
# BASELINE configs[2..4]: the same JSON vocabulary for the three large problems
CONFIGS = {
    # name: (module, builder, data fn, instances per GPU, default variant, CPU sample, workload label)
    "l1l2": ("tenscalc_amd.problems.l1l2", "l1l2", "l1l2_data", 16384, "interp8", 2048,
             "l1l2estimationCS N=200 (1792x1792 KKT, 54 k instructions), batch=%d per GPU, fp64 (BASELINE.json configs[2])"),
    "dcmotor": ("tenscalc_amd.problems.mpc_dcmotor", "mpc_dcmotor", "dcmotor_data", 32768, "strict", 32768,
                "mpc_dcmotor T=30 (323x323 KKT), batch=%d parameter sweep per GPU, fp64 (BASELINE.json configs[3]: the "
                "whole 8-GPU sweep of 32768 on one GPU by default)"),
    "quadcopter": ("tenscalc_amd.problems.mpc_quadcopter", "mpc_quadcopter", "quadcopter_data", 512, "ldsm8", 512,
                   "mpc_quadcopter T=100 (1406x1406 KKT, 67 k instructions), batch=%d per GPU, fp64 (BASELINE.json "
                   "configs[4]: the per-GPU share of 4096 over 8 GPUs)"),
}


def bench_config(args):
    """python bench.py --config l1l2|dcmotor|quadcopter [--variant V] [--batch B]: one step = one batched solve of B
    instances per GPU with the inputs resident in HBM; same timing contract and JSON fields as the sls line."""
    import importlib
    import numpy as np
    import torch
    import torch.distributed as dist
    import __graft_entry__ as g
    from tenscalc_amd.runtime import HipSolver
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)
    modname, mk, datafn, B0, var0, ncpu, label = CONFIGS[args.config]
    mod = importlib.import_module(modname)
    SOLVE = mod.SOLVE_ARGS
    variant = args.variant or var0                       # any key of __graft_entry__.VARIANT_KW (interp8, ldsm4, strict, ...)
    if variant not in g.VARIANT_KW:
        raise SystemExit("bench.py: --variant for --config %s must be one of: %s" % (args.config, ", ".join(sorted(g.VARIANT_KW))))
    if args.config == "dcmotor" and not args.variant and 0 < args.batch <= 8192:
        variant = "w32x"      # one GPU's share of the 8-GPU sweep: the wave-cooperative lowering (LDS-resident state) wins
    B = args.batch or B0
    spec = getattr(mod, mk)(folder=g.GENERATED)
    kw = dict(g.VARIANT_KW[variant])
    if rank == 0:
        spec.build(**kw)
    if world > 1:
        dist.barrier(device_ids=[local])
    so = spec.build(**kw)
    names = [n for n, _ in spec.parameters + spec.variables]
    nd = min(B, 256)                                           # 256 distinct seeded instances, tiled over the batch
    ds = [getattr(mod, datafn)(rank * 256 + k) for k in range(nd)]
    stacked = {n: np.stack([np.asarray(ds[k % nd][n], dtype=float).reshape(-1, order="F") for k in range(B)]) for n in names}
    P = max(1, int(args.pipeline))
    objs = [HipSolver(so, B=B, device=local) for _ in range(P)]
    for o_ in objs:
        for n in names:
            o_.set(n, stacked[n])
    obj = objs[0]
    streams = [torch.cuda.current_stream()] + [torch.cuda.Stream(device=dev) for _ in range(P - 1)]
    ae = SOLVE.get("addEye2Hessian", (1e-9, 1e-9))
    for k in range(args.warmup * P):
        objs[k % P].solve_async(SOLVE["mu0"], SOLVE["maxIter"], ae, stream=streams[k % P].cuda_stream)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    if world > 1:      # one untimed gather of the same shape: RCCL sets up its channels at the first collective
        nout_ = sum(o["numel"] for o in obj.outputs)
        w_ = torch.zeros((nout_ + 2, B), dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(torch.empty((world * (nout_ + 2), B), dtype=torch.float64, device=dev), w_)
        dist.barrier(device_ids=[local])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        s_ = streams[k % P]
        evs[k][0].record(s_)
        objs[k % P].solve_async(SOLVE["mu0"], SOLVE["maxIter"], ae, stream=s_.cuda_stream)
        evs[k][1].record(s_)
    for s_ in streams[1:]:
        torch.cuda.current_stream().wait_stream(s_)
    optr, sptr, iptr = obj.device_views()
    nout = sum(o["numel"] for o in obj.outputs)
    ST = torch.as_tensor(_DevArray(sptr, (B,), "<i4"), device=dev)
    IT = torch.as_tensor(_DevArray(iptr, (B,), "<i4"), device=dev)
    if world > 1:      # the path's only exchange: one gather of status / iteration counts / outputs after the K steps
        O = torch.as_tensor(_DevArray(optr, (nout, B), "<f8"), device=dev)
        from tenscalc_amd.sharding import pack_results
        packed = pack_results(O, ST, IT)
        gathered = torch.empty((world,) + tuple(packed.shape), dtype=packed.dtype, device=dev)
        dist.all_gather_into_tensor(gathered.view(world * packed.shape[0], B), packed)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier(device_ids=[local])
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    ok = (ST == 0).sum().to(torch.float64).reshape(1)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(ok, op=dist.ReduceOp.SUM)
    elapsed = float(el.item())
    if rank == 0:
        it = IT.cpu().numpy()
        st = ST.cpu().numpy()
        kavg = float(np.mean([a.elapsed_time(b_) for a, b_ in evs])) * 1e-3
        nin = sum(int(np.prod(s_)) if len(s_) else 1 for _, s_ in spec.parameters + spec.variables)
        algo = 8 * (nin + nout) + 8                               # inputs + outputs + (status, iter)
        achieved = algo * B / kavg / 1e9
        out = {"metric": "IPM solves/sec (whole node), %s batch" % args.config, "value": float(ok.item()) * args.steps / elapsed,
               "unit": "solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f64", "data": "synthetic",
               "config": {"workload": label % B, "batch_per_gpu": B, "mu0": SOLVE["mu0"], "maxIter": SOLVE["maxIter"],
                          "variant": variant, "parity_class": g.PARITY_CLASS.get(variant, "solver-tolerance"),
                          "pipeline": P,
                          "solved_fraction": float(ok.item()) / (B * world), "iter_median": float(np.median(it)),
                          "iter_max": int(it.max()),
                          "sharding": "independent instances per rank; one RCCL all_gather of the results after the K steps"},
               "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                            "frac": achieved / HBM_PEAK_GBPS, "traffic": measured_traffic(args.config + "_" + variant, B),
                            "kernel": "tc_solve_kernel", "kernel_ms_avg": kavg * 1e3,
                            "algorithmic_bytes_per_launch": algo * B, "algorithmic_bytes_per_solve": algo}}
        if args.config == "quadcopter":      # its tape calls libm pow(): device and glibc differ by 1 ulp on 26 % of arguments
            out["config"]["parity_note"] = ("the class applies between the device back-ends and to the oracle built with the "
                                            "same libm; vs the cmex reference 1-iteration agreement 3e-10, "
                                            "tests/test_gpu_interp.py")
        if not args.no_cpu and world == 1:
            out["cpu_baseline"] = cpu_baseline_config(spec, args.config, stacked, names, min(B, ncpu), SOLVE, st, it)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier(device_ids=[local])
        dist.destroy_process_group()
    for o_ in objs:
        o_.close()


def cpu_baseline_config(spec, which, stacked, names, n, SOLVE, st, it):
    """bounded sample of the same workload on one host core: the generated-C twin where it is prebuilt (dcmotor),
    else the oracle's tape interpreter (the twins of the big tapes are 44 MB C files gcc needs an hour for)"""
    import numpy as np
    sub = {k: v[:n] for k, v in stacked.items()}
    twin = None
    if which == "dcmotor":
        try:
            from oracle.c_emitter import CpuTwin
            twin = CpuTwin(spec)
        except (FileNotFoundError, OSError):
            twin = None
    if twin is not None:
        run = lambda: twin.solve_batch(sub, n, mu0=SOLVE["mu0"], maxIter=SOLVE["maxIter"])      # noqa: E731
        how = "generated-C twin + reference ipmPD_CSsolver.c, gcc -O1"
    else:
        from tenscalc_amd.tapeio import save_tape
        from oracle.tc_oracle import Oracle
        import tempfile
        tape_path = os.path.join(tempfile.mkdtemp(prefix="tc_bench_"), which + ".tape")
        save_tape(spec.tape, tape_path)
        c = spec.classname
        o = Oracle(tape_path)
        run = lambda: o.solve_batch({f"{c}_set_{k}": v for k, v in sub.items()}, c + "_getOutputs", n,      # noqa: E731
                                    mu0=SOLVE["mu0"], maxIter=SOLVE["maxIter"])
        how = "oracle tape interpreter (oracle/tc_oracle.c: same arithmetic as the generated C, about 2-5x slower)"
    t0 = time.perf_counter()
    reps = 0
    while reps == 0 or (time.perf_counter() - t0 < 8.0 and reps < 64):        # ~10 s of CPU work
        out, st2, it2 = run()
        reps += 1
    dt = (time.perf_counter() - t0) / reps
    return {"value": float((st2 == 0).sum()) / dt, "unit": "solves/s", "cores": 1, "kind": "port",
            "sample": "first %d instances of the benchmark batch, %s, sequential on 1 of %d host cores, %.2f s per pass"
                      % (n, how, os.cpu_count(), dt),
            "same_status_iter": bool(np.array_equal(st[:n], st2) and np.array_equal(it[:n], it2))}


if __name__ == "__main__":
    main()
