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
    ap.add_argument("--batch", type=int, default=65536, help="instances per GPU")
    ap.add_argument("--variant", default=os.environ.get("TC_BENCH_VARIANT", "w8x"), choices=["strict", "fast", "w8", "w8h", "regs", "regsl", "w8s", "w8x"],
                    help="w8x (default): 8 lanes/instance wave-cooperative lowering of the REFERENCE tape with the "
                         "reference's arithmetic kept bit for bit (parity class bit-exact: status, iteration count and "
                         "outputs identical to the cmex reference on every instance); strict: 1 thread/instance, "
                         "-ffp-contract=off, bit-exact too; fast: same with FMA; w8: the wave-cooperative lowering with "
                         "FMA, reassociated reductions and fast division; w8s: the same of the hoisted tape (parameter-only "
                         "contractions first, tenscalc_amd/hoist.py), inputs streamed, LDS slots shared, 6 wavefronts per "
                         "CU -- parity class solver-tolerance; w8h / regs / regsl: other lowerings of the hoisted tape")
    ap.add_argument("--cpu-sample", type=int, default=int(os.environ.get("TC_CPU_SAMPLE", "65536")))
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

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

    B = args.batch
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
    from tenscalc_amd.sharding import pack_results
    # the path's only exchange: results (J, x, status, iter) of every rank gathered on every rank.  It overlaps
    # the next step's solve: the results are packed on the launch stream into one of two buffers, the collective
    # is enqueued asynchronously (RCCL runs it on its own stream after the pack) and is only waited for when its
    # buffer comes up for reuse -- and, for all of them, before the timed region ends.
    gathered = [torch.empty((world, 11, B), dtype=torch.float64, device=cdev) for _ in range(2)] if world > 1 else None
    packed = [None, None]
    pending = [None, None]
    nex = [0]

    def exchange():
        k = nex[0] & 1
        nex[0] += 1
        if pending[k] is not None:
            pending[k].wait()
        packed[k] = pack_results(O, ST, IT).to(cdev).contiguous()
        pending[k] = dist.all_gather_into_tensor(gathered[k].view(world * 11, B), packed[k], async_op=True)

    def drain():
        for k in range(2):
            if pending[k] is not None:
                pending[k].wait()
                pending[k] = None

    def step():
        obj.solve_async(SOLVE_ARGS["mu0"], SOLVE_ARGS["maxIter"], SOLVE_ARGS["addEye2Hessian"], stream=stream)
        if world > 1:
            exchange()

    for _ in range(args.warmup):
        step()
    if world > 1:
        drain()
    # --- timed region: barrier + synchronize on both sides, exactly K steps
    kern_ms = []
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    if world > 1:
        bar()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        evs[k][0].record()
        obj.solve_async(SOLVE_ARGS["mu0"], SOLVE_ARGS["maxIter"], SOLVE_ARGS["addEye2Hessian"], stream=stream)
        evs[k][1].record()
        if world > 1:
            exchange()
    if world > 1:
        drain()                    # every gather of the K steps has completed inside the timed region
    torch.cuda.synchronize()
    if world > 1:
        bar()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kern_ms = [a.elapsed_time(b_) for a, b_ in evs]          # events on the launch stream
    el = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
    ok = (ST == 0).sum().to(torch.float64).reshape(1).to(cdev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(ok, op=dist.ReduceOp.SUM)
    elapsed = float(el.item())
    solved_per_step = float(ok.item())                        # only status==0 instances count
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
                       "sharding": "independent instances per rank, RCCL all_gather of results only (overlapped with the next solve)"},
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


def cpu_baseline(spec, A, b, x0, sample, solve_args):
    """The CPU restatement of the reference path timed on ONE host core over the first `sample`
    instances of the same workload (kind "port").  Preferred: the generated-C twin (oracle/c_emitter.py:
    the tape as reference-style straight-line C + the reference's ipmPD_CSsolver.c, gcc -O1, prebuilt
    in the development container); fallback: the tape interpreter oracle/tc_oracle.c."""
    import numpy as np
    import __graft_entry__ as g
    c = spec.classname
    n = min(sample, A.shape[1])
    Ah = np.ascontiguousarray(A[:, :n].t().cpu().numpy())
    bh = np.ascontiguousarray(b[:, :n].t().cpu().numpy())
    xh = np.ascontiguousarray(x0[:, :n].t().cpu().numpy())
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
    except (FileNotFoundError, OSError):
        from oracle.tc_oracle import Oracle
        o = Oracle(os.path.join(g.GENERATED, "@" + c, c + ".tape"))
        sets = lambda k: {c + "_set_A": Ah[:k], c + "_set_b": bh[:k], c + "_set_x": xh[:k]}  # noqa: E731
        o.solve_batch(sets(64), c + "_getOutputs", 64, mu0=solve_args["mu0"], maxIter=solve_args["maxIter"])
        t0 = time.perf_counter()
        out, st, it = o.solve_batch(sets(n), c + "_getOutputs", n, mu0=solve_args["mu0"], maxIter=solve_args["maxIter"])
        dt = time.perf_counter() - t0
        what = "oracle/tc_oracle.c (tape interpreter + restated ipmPD_CSsolver loop, gcc -O1)"
    return {"value": float((st == 0).sum()) / dt, "unit": "solves/s", "cores": 1, "kind": "port",
            "sample": "first %d instances of the benchmark batch, %s, sequential on 1 of %d host cores, %.2f s per pass"
                      % (n, what, os.cpu_count(), dt),
            "us_per_solve": dt / n * 1e6}


if __name__ == "__main__":
    main()
