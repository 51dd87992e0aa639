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


_POW_CHECK = {}


def parity_fields(label, tape, defines):
    """config.parity_class, checked: a "bit-exact" label is printed only if the device's pow replacements return this host's
    pow on 2e5 arguments -- once per process, on the GPU when there is one (tenscalc_amd/tools/powcheck.py); the outcome
    rides along as config.libm_check.  TC_BENCH_STANDIN rehearsals have no device and check the host build."""
    from tenscalc_amd.tools.powcheck import checked_parity_class, pow_parity, tape_calls_pow
    if label == "bit-exact" and any(tape_calls_pow(tape, defines)) and "r" not in _POW_CHECK:
        _POW_CHECK["r"] = pow_parity(device=False if os.environ.get("TC_BENCH_STANDIN") else None)
    label, chk = checked_parity_class(label, tape, defines, check=_POW_CHECK.get("r"))
    out = {"parity_class": label}
    if chk is not None:
        out["libm_check"] = chk
    return out


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


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch(args, argv):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment: start N FRESH rank processes (one per
    GPU) through torch.distributed.run and relay rank 0's JSON line.  This parent never touches HIP (no torch.cuda call,
    no exec of a process that has): the ranks are children, and the parent exits with the launcher's code."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: what RCCL needs on this pool
    env.setdefault("OMP_NUM_THREADS", "1")
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    for ln in r.stdout.splitlines():
        if ln not in lines:
            print(ln, file=sys.stderr)
    if lines:
        print(lines[-1], flush=True)
    return r.returncode or (0 if lines else 1)


class Ranks:
    """rank / world / device of this process and the process group (RCCL = backend "nccl"; gloo for the rehearsal
    modes).  TC_BENCH_SINGLE_DEVICE=1: several ranks share GPU 0, collectives through gloo on host copies (RCCL refuses
    two ranks on one device).  TC_BENCH_STANDIN=1: no GPU at all -- the solver is replaced by a stand-in that produces
    recognisable results on the CPU, so that launcher, sharding, exchange and reduction can be run end to end where
    there is no GPU (tests/test_bench_launcher.py); its line is marked and never a measurement."""

    def __init__(self, args, ap):
        import torch
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local = int(os.environ.get("LOCAL_RANK", "0"))
        self.standin = bool(os.environ.get("TC_BENCH_STANDIN"))
        self.single = bool(os.environ.get("TC_BENCH_SINGLE_DEVICE")) or self.standin
        if self.world != args.gpus:
            ap.error("--gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, self.world))
        if self.single:
            self.local = 0
        if self.standin:
            self.dev = self.cdev = torch.device("cpu")
        else:
            torch.cuda.set_device(self.local)
            self.dev = torch.device("cuda", self.local)
            self.cdev = torch.device("cpu") if self.single else self.dev
        self.dist = None
        # TC_BENCH_FORCE_DIST=1: form the process group even for ONE rank, so that the RCCL branch (init with device_id,
        # barrier(device_ids), all_gather_into_tensor, all_reduce) runs on a one-GPU box (tests/test_gpu_bench_nccl.py)
        if self.world > 1 or os.environ.get("TC_BENCH_FORCE_DIST"):
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if self.single:
                dist.init_process_group(backend="gloo")
            else:
                dist.init_process_group(backend="nccl", device_id=self.dev)   # RCCL on ROCm; one rank per GPU
            self.dist = dist
            if dist.get_world_size() != self.world:
                raise SystemExit("bench.py: process group has %d ranks, expected %d" % (dist.get_world_size(), self.world))
        self.devices = self._device_ids()

    @property
    def multi(self):
        """a process group exists: the exchange / reduction code runs (world > 1, or forced for a single rank)"""
        return self.dist is not None

    def barrier(self):
        if self.dist is not None:
            if self.single:
                self.dist.barrier()
            else:
                self.dist.barrier(device_ids=[self.local])

    def sync(self):
        if not self.standin:
            import torch
            torch.cuda.synchronize()

    def _device_ids(self):
        """identity of every rank's device, gathered over the process group: n_gpus is what the group SAW, not an echo
        of WORLD_SIZE"""
        import socket
        if self.standin:
            me = "standin:%d" % self.rank
        else:
            import torch
            pr = torch.cuda.get_device_properties(self.local)
            ident = getattr(pr, "uuid", None) or getattr(pr, "pci_bus_id", None) or self.local
            me = "%s:%s" % (socket.gethostname(), ident)
        if self.dist is None:
            return [me]
        got = [None] * self.world
        self.dist.all_gather_object(got, me)
        if not self.single and len(set(got)) != self.world:
            raise SystemExit("bench.py: %d ranks but only %d distinct devices: %s" % (self.world, len(set(got)), got))
        return got

    def finish(self):
        if self.dist is not None:
            self.barrier()
            self.dist.destroy_process_group()


class _StandInSolver:
    """TC_BENCH_STANDIN only: plays the per-rank solver on the CPU.  Not a solver and not the oracle: `solve` marks every
    instance whose first input is finite as solved and writes a checksum of its inputs, so that a test can tell which
    rank produced which column."""

    def __init__(self, B, nout):
        import torch
        self.B, self.nout = B, nout
        self.O = torch.zeros((nout, B), dtype=torch.float64)
        self.ST = torch.full((B,), -1, dtype=torch.int32)
        self.IT = torch.zeros((B,), dtype=torch.int32)
        self.inputs = {}

    def set_device(self, name, t, *a, **k):
        self.inputs[name] = t

    def solve_async(self, *a, **k):
        import torch
        acc = sum(v.reshape(-1, self.B).sum(0) for v in self.inputs.values())
        self.O[:] = acc
        self.ST[:] = torch.where(torch.isfinite(acc), 0, 4).to(torch.int32)
        self.IT[:] = 11

    def close(self):
        pass


class _StandInConfigSolver(_StandInSolver):
    """the same for `bench_config`: the handle surface that function uses (set from host arrays, outputs, views)"""

    def __init__(self, B, spec):
        self.outputs = [{"name": n_, "numel": int(__import__("numpy").prod(sz)) if len(sz) else 1} for n_, sz in spec.outputs]
        super().__init__(B, sum(o["numel"] for o in self.outputs))

    def set(self, name, arr):
        import torch
        self.inputs[name] = torch.as_tensor(arr, dtype=torch.float64).reshape(self.B, -1).T.contiguous()     # [numel][B]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=0, help="instances per GPU (default: the config's); weak scaling")
    ap.add_argument("--total", type=int, default=0,
                    help="STRONG scaling: this many instances in all, contiguous ranges [r*T/N, (r+1)*T/N) per rank "
                         "(tenscalc_amd/sharding.py:shard_range) -- BASELINE configs[3] (32768 over 8 GPUs) and configs[4] "
                         "(4096 over 8 GPUs) are stated as totals")
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
    ap.add_argument("--no-calib", action="store_true",
                    help="--config with an lds* variant: skip the extra solve of <= 256 instances that measures the time of one "
                         "Newton iteration of one instance (profiling runs: every launch of the kernel is then a full batch)")
    ap.add_argument("--no-configs", action="store_true",
                    help="default N=1 sls run: do not attach the short runs of BASELINE configs[2..4] under \"configs\"")
    ap.add_argument("--per-step-events", action="store_true",
                    help="sls line: one HIP event pair per launch instead of one pair around the K launches")
    ap.add_argument("--sustain", type=float, default=2.0,
                    help="seconds of back-to-back steps after the timed region for `sustained_ms_per_step` (0 = skip)")
    ap.add_argument("--config", default="sls", choices=["sls", "l1l2", "dcmotor", "quadcopter"],
                    help="sls (default): BASELINE.json's metric, configs[1].  The others are BASELINE configs[2..4] in the "
                         "same JSON vocabulary: l1l2 16384 instances, dcmotor 32768 (the whole 8-GPU sweep on one GPU), "
                         "quadcopter 4096 (configs[4]; --batch 512 = its per-GPU share over 8 GPUs) per GPU; --batch / --total override")
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
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch(args, sys.argv[1:]))          # before anything touches the GPU
    if args.batch and args.total:
        ap.error("--batch (per GPU, weak scaling) and --total (strong scaling) exclude each other")
    import __graft_entry__ as g
    if args.config == "sls":
        args.variant = args.variant or g.BENCH_DEFAULT
        sls_variants = [v for v, kw in g.VARIANT_KW.items() if kw.get("body", "code") == "code" and v not in g.CONFIG_ONLY]
        if args.variant not in sls_variants:
            ap.error("--variant for --config sls: " + ", ".join(sls_variants))
    elif args.variant and args.variant not in g.VARIANT_KW:
        ap.error("--variant for --config %s must be one of: %s" % (args.config, ", ".join(sorted(g.VARIANT_KW))))
    R = Ranks(args, ap)
    if args.config == "sls":
        out = bench_sls(args, R)
        if (R.rank == 0 and R.world == 1 and not args.no_configs and not R.standin and not args.batch and not args.total
                and args.variant == g.BENCH_DEFAULT):
            # BASELINE configs[2..4] ride along in the driver-timed invocation: 3 steps each, bounded CPU samples
            out["configs"] = {}
            for name in ("l1l2", "dcmotor", "quadcopter", "quadcopter_share512"):
                t0 = time.perf_counter()
                try:
                    sub = argparse.Namespace(**dict(vars(args), config=name.split("_")[0], variant="", steps=3, warmup=1,
                                                    pipeline=1, cpu_budget_s=4.0))
                    if name == "quadcopter_share512":        # one GPU's share of configs[4] over 8 GPUs; CPU figure: see above
                        sub.batch, sub.no_cpu = 512, True
                    out["configs"][name] = bench_config(sub, R)
                except Exception as e:        # a side line must not take the headline down with it
                    out["configs"][name] = {"error": "%s: %s" % (type(e).__name__, e)}
                out["configs"][name]["wall_s"] = time.perf_counter() - t0
    else:
        out = bench_config(args, R)
    if R.rank == 0:
        print(json.dumps(out), flush=True)
    R.finish()


def bench_sls(args, R):
    """BASELINE configs[1]: one step = one tc_solve_kernel launch over this rank's sls instances, inputs resident in
    HBM.  Returns rank 0's JSON object (None on the other ranks)."""
    import numpy as np
    import torch
    import __graft_entry__ as g
    from tenscalc_amd.problems.sls import SOLVE_ARGS
    from tenscalc_amd.sharding import ResultExchange, pack_results, reduce_timing, run_timed, shard_range
    rank, world, dev = R.rank, R.world, R.dev
    spec = g.sls_spec(args.variant)
    kw = dict(g.VARIANT_KW[args.variant])
    if os.environ.get("TC_EXPERIMENT"):          # e.g. TC_EXPERIMENT="min_waves=2,block=256" (tuning runs)
        for kv in os.environ["TC_EXPERIMENT"].split(","):
            k_, v_ = kv.split("=")
            kw[k_] = int(v_)
        kw["tag"] = (kw.get("tag", "fast") + "_" + os.environ["TC_EXPERIMENT"].replace("=", "").replace(",", "_"))
    if args.total:
        lo, hi = shard_range(args.total, rank, world)
        B, Bmax = hi - lo, -(-args.total // world)
    else:
        B = Bmax = args.batch or 65536
    if R.standin:
        gen = torch.Generator().manual_seed(1234 + rank)
        A, b, x0 = (torch.rand((n_, B), generator=gen, dtype=torch.float64) for n_ in (800, 100, 8))
        obj = _StandInSolver(B, 9)
        O, ST, IT, stream = obj.O, obj.ST, obj.IT, None
        for nm, t in (("A", A), ("b", b), ("x", x0)):
            obj.set_device(nm, t)
    else:
        from tenscalc_amd.runtime import LAYOUT_BATCH_MINOR, HipSolver
        if rank == 0:
            so = spec.build(**kw)          # prebuilt in-tree normally; compiles here only if missing
        R.barrier()
        so = spec.build(**kw)
        A, b, x0 = make_batch(B, rank, dev)
        obj = HipSolver(so, B=B, device=R.local)
        # everything of the timed region -- launches, timing events, result packing -- goes on the HANDLE's own stream
        # (made torch's current stream): a foreign stream costs an event record + wait on either side of every launch
        # (include/tenscalc_hip.h "Devices and streams"), 0.03 ms of a 1.15 ms step
        torch.cuda.synchronize()
        hstream = torch.cuda.ExternalStream(obj.stream(), device=dev)
        torch.cuda.set_stream(hstream)
        stream = hstream.cuda_stream
        for nm, t in (("A", A), ("b", b), ("x", x0)):
            obj.set_device(nm, t, LAYOUT_BATCH_MINOR, stream=stream)
        torch.cuda.synchronize()
        optr, sptr, iptr = obj.device_views()
        O = torch.as_tensor(_DevArray(optr, (9, B), "<f8"), device=dev)
        ST = torch.as_tensor(_DevArray(sptr, (B,), "<i4"), device=dev)
        IT = torch.as_tensor(_DevArray(iptr, (B,), "<i4"), device=dev)
    # the path's only exchange: results (J, x, status, iter) of every rank gathered on every rank -- ONE gather after
    # the K steps by default; with --gather-every-step after every step, overlapped with the next
    # step's solve (packed on the launch stream, the collective asynchronous on RCCL's own stream).  The control
    # flow is tenscalc_amd/sharding.py:run_timed, rehearsed with gloo on the CPU by tests/test_sharding_gloo.py and
    # end to end by tests/test_bench_launcher.py.
    multi = R.multi
    wflag = world if world > 1 else (2 if multi else 1)      # run_timed / reduce_timing only ask "> 1?"
    every = multi and args.gather_every_step
    xch = ResultExchange(world, (11, Bmax), R.cdev) if multi else None

    def packed():
        p = pack_results(O, ST, IT)
        if B < Bmax:        # unequal shards of a --total run: pad to the common width, status -1 = no instance
            pad = torch.zeros((11, Bmax - B), dtype=p.dtype, device=p.device)
            pad[9] = -1.0
            p = torch.cat([p, pad], 1)
        return p

    def solve(_k=0):
        obj.solve_async(SOLVE_ARGS["mu0"], SOLVE_ARGS["maxIter"], SOLVE_ARGS["addEye2Hessian"], stream=stream)

    for _ in range(args.warmup):
        solve()
        if every:
            xch.exchange(packed())
    if multi:
        if not every:      # one untimed gather: RCCL sets up its channels at the first collective of a kind and size
            xch.exchange(packed())
        xch.drain()
    # --- timed region: barrier + synchronize on both sides, exactly K steps
    # kernel time: HIP events on the launch stream over the timed region.  ONE pair around the K launches by default
    # (average launch duration = span / K, the gaps between launches included): a pair per launch puts two more
    # packets between consecutive kernels and costs 2-3 % of the step (measured: 1.19 vs 1.14 ms); --per-step-events
    # restores it (then kernel_ms_avg excludes the gaps)
    per_step = bool(args.per_step_events)
    nev = args.steps if per_step else 1
    evs = [] if R.standin else [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                                for _ in range(nev)]

    def step(k):
        if evs and (per_step or k == 0):
            evs[k if per_step else 0][0].record()
        solve()
        if evs and (per_step or k == args.steps - 1):
            evs[k if per_step else 0][1].record()
    elapsed = run_timed(args.steps, step, packed, wflag, every=every, barrier=R.barrier, sync=R.sync, exchange=xch)
    if evs:          # events on the launch stream
        kern_ms = [a.elapsed_time(b_) / (1 if per_step else args.steps) for a, b_ in evs]
    else:
        kern_ms = [elapsed / args.steps * 1e3]
    elapsed, solved_per_step = reduce_timing(elapsed, int((ST == 0).sum().item()), wflag, R.cdev)   # only status==0 count
    value = solved_per_step * args.steps / elapsed
    gathered_ok = None
    if multi:       # what the exchange delivered: every rank's status row, instances counted on rank 0
        last = xch.last()
        gathered_ok = int((last[:, 9, :] == 0).sum().item())
    sustained = None
    if args.sustain > 0 and world == 1 and not R.standin:
        # the contract's 20 steps are 24 ms: a second figure over >= args.sustain seconds of back-to-back launches says
        # whether the clocks hold (same kernel, same inputs, nothing else changed)
        n_s, t_s = 0, time.perf_counter()
        while time.perf_counter() - t_s < args.sustain:
            for _ in range(50):
                solve()
            torch.cuda.synchronize()
            n_s += 50
        sustained = (time.perf_counter() - t_s) / n_s * 1e3
    out = None
    if rank == 0:
        it = IT.cpu().numpy()
        kavg = float(np.mean(kern_ms)) * 1e-3
        achieved = ALGO_BYTES_PER_SOLVE * B / kavg / 1e9
        total = args.total or B * world
        out = {
            "metric": "IPM solves/sec (whole node), sls N=100 n=8 batch",
            "value": value, "unit": "solves/s", "n_gpus": len(set(R.devices)), "ranks": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if args.total else "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": ("sls box-constrained LS N=100 n=8, %s random (A,b), fp64 (BASELINE.json configs[1])"
                                    % (("batch=%d in all, sharded over %d GPU(s)" % (args.total, world)) if args.total
                                       else "batch=%d per GPU" % B)),
                       "batch_per_gpu": B, "batch_total": total, "mu0": SOLVE_ARGS["mu0"], "maxIter": SOLVE_ARGS["maxIter"],
                       "variant": args.variant,
                       **parity_fields(g.PARITY_CLASS.get(args.variant, "solver-tolerance"), spec.tape, spec.defines),
                       "tape": ("hoisted (parameter-only contractions first); results within the stated tolerance of the "
                                "reference tape, tests/test_gpu_parity.py" if args.variant in g.HOISTED else "reference"),
                       "solved_fraction": solved_per_step / total,
                       "iter_median": float(np.median(it)), "iter_max": int(it.max()),
                       "devices": R.devices,
                       "sharding": "independent instances per rank; RCCL all_gather of the results only, "
                                   + ("after every step, overlapped with the next solve" if every else "once after the K steps")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": measured_traffic(args.variant, B),
                         "traffic_from": traffic_from(args.variant, B),
                         "kernel": "tc_solve_kernel", "kernel_ms_avg": kavg * 1e3,
                         "kernel_ms_from": ("HIP event pair per launch on the launch stream" if per_step else
                                            "one HIP event pair around the K launches on the launch stream: span / K"),
                         "algorithmic_bytes_per_launch": ALGO_BYTES_PER_SOLVE * B,
                         "algorithmic_bytes_per_solve": ALGO_BYTES_PER_SOLVE,
                         # what actually binds this kernel: dependent fp64 chains at one wavefront per
                         # SIMD with the per-instance state on chip -- neither HBM nor VALU throughput; "bound" stays
                         # "hbm" because that is the roofline north_star prices the path against
                         "binds": "fp64 dependent-issue latency + LDS round trips at 1 wavefront/SIMD (on-chip state); "
                                  "HBM traffic is 1.05x algorithmic",
                         # this path is sparse scalar fp64 work, so the vector-fp64 fraction is
                         # reported beside the HBM fraction the north star asks for
                         "fp64_valu": None if R.standin else tape_flops(spec, it, B, kavg)},
        }
        if out["roofline"]["fp64_valu"]:
            out["roofline"]["lane_redundancy"] = lane_redundancy(args.variant, B, out["roofline"]["fp64_valu"]["flops_per_solve"])
        if sustained is not None:
            out["sustained_ms_per_step"] = sustained
            out["sustained_value"] = float((ST == 0).sum().item()) / (sustained * 1e-3)
        if gathered_ok is not None:
            out["config"]["gathered_solved_last_step"] = gathered_ok
        if R.standin:
            out["data"] = "STAND-IN: no solver ran (TC_BENCH_STANDIN rehearsal of the multi-rank control flow)"
            out["standin"] = True
        if not args.no_cpu and world == 1 and not R.standin:      # CPU baseline: rank 0 at N=1 only
            # always the reference's own tape (never the hoisted one): that is the path the reference runs
            out["cpu_baseline"] = cpu_baseline(g.sls_spec("strict"), A, b, x0, args.cpu_sample, SOLVE_ARGS)
    if not R.standin:
        torch.cuda.synchronize()
        torch.cuda.set_stream(torch.cuda.default_stream(dev))      # the handle's stream dies with the handle
    obj.close()
    return out


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


def lane_redundancy(variant, B, flops_per_solve):
    """lane-operations the kernel issues per useful tape operation: SQ_INSTS_VALU (wave-level VALU instructions per launch,
    from the committed PMC pass) x 64 lanes / (tape flops per solve x B).  sls / w8x: 3.5 -- the sequential core of an
    instance is computed redundantly by its 8 lanes.  None without a counter run for this variant and batch."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            e = json.load(f).get("%s_B%d" % (variant, B))
        return e["sq_insts_valu"] * 64.0 / (flops_per_solve * B) if e and e.get("sq_insts_valu") else None
    except (OSError, ValueError, KeyError, ZeroDivisionError):
        return None


def iteration_spread(it, variant, B, kernel_s, calib=None):
    """How much of a batch is its tail.  `iter_hist`: Newton iterations per instance in 8 bins; `straggler_floor_ms`: the
    slowest instance ALONE -- iter_max x the time one Newton iteration of one instance takes on this back-end; the batch
    cannot end before its slowest instance does, wherever the dispatcher started it.  For the back-ends that give an
    instance a workgroup of its own (`lds*`) the per-iteration time is MEASURED in this run: `calib` = (seconds, iterations
    of the slowest instance) of one more solve of the first <= 256 instances, each alone on its CU, whose time is that
    instance's.  `throughput_floor_ms` = all iterations of the batch at that rate over the CUs (x residents per CU where the
    class admits several).  For the tile back-ends an instance's iterations are not separable from its tile's: the floor
    is iter_max x kernel / mean(tile maxima), and `lockstep_waste` the share of a tile's walk spent on finished lanes."""
    import numpy as np
    it = np.asarray(it, dtype=np.int64)
    lo, hi = int(it.min()), int(it.max())
    edges = np.unique(np.round(np.geomspace(max(lo, 1), max(hi, 2) + 1, 9)).astype(np.int64))
    hist, edges = np.histogram(it, bins=edges)
    out = {"iter_mean": float(it.mean()), "iter_total": int(it.sum()),
           "iter_hist": {"edges": [int(e) for e in edges], "counts": [int(c) for c in hist]}}
    if variant.startswith("lds"):
        if calib:
            CUS = 256
            t_iter = calib[0] / max(int(calib[1]), 1)                  # seconds per Newton iteration of one instance
            out["ms_per_iteration_one_instance"] = t_iter * 1e3
            out["ms_per_iteration_from"] = "one solve of the first %d instances (one per CU): %.2f ms / %d iterations of its slowest" % (
                calib[2], calib[0] * 1e3, calib[1])
            out["straggler_floor_ms"] = hi * t_iter * 1e3
            res = calib[3] if len(calib) > 3 else 1            # workgroups of this class that fit a CU's LDS side by side
            out["residents_per_cu"] = res
            out["throughput_floor_ms"] = float(it.sum()) * t_iter * 1e3 / min(CUS * res, B)
            out["straggler_position"] = float(int(np.argmax(it))) / max(len(it) - 1, 1)     # 0 = dispatched first, 1 = last
            out["batch_over_floor"] = kernel_s * 1e3 / max(out["straggler_floor_ms"], out["throughput_floor_ms"])
            out["tail_note"] = ("one instance per workgroup: batch >= max(straggler_floor_ms, throughput_floor_ms); the straggler "
                                "starts when the workgroup dispatcher reaches it, which no launch order can know in advance")
    else:
        tiles = it[:len(it) // 64 * 64].reshape(-1, 64).max(axis=1) if len(it) >= 64 else it[None].max(axis=1)
        out["tile_iter_max_mean"] = float(tiles.mean())
        out["straggler_floor_ms"] = kernel_s * 1e3 * hi / float(tiles.mean())
        out["lockstep_waste"] = 1.0 - float(it.mean()) / float(tiles.mean())       # share of a tile's walk spent on finished lanes
    return out


def traffic_from(variant, B):
    """where `roofline.traffic` / `lane_redundancy` of this line come from: they are REPLAYED from a committed counter pass
    (profiles/traffic.json, builder-run rocprofv3 --pmc passes of this same command on another box, possibly in an earlier
    round), not measured during this run -- the JSON says so itself"""
    key = "%s_B%d" % (variant, B)
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            e = json.load(f).get(key)
    except (OSError, ValueError):
        e = None
    if not e:
        return None
    return {"file": "profiles/traffic.json", "entry": key, "round": e.get("round"),
            "how": "replayed: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (and SQ_INSTS_VALU) passes of this command, "
                   "run by the builder on another MI355X box in that round; NOT a measurement of this run",
            "kernel_ms_of_that_run": e.get("kernel_avg_ns", 0) * 1e-6 or None}


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
    "dcmotor": ("tenscalc_amd.problems.mpc_dcmotor", "mpc_dcmotor", "dcmotor_data", 32768, "ldscs", 32768,
                "mpc_dcmotor T=30 (323x323 KKT), batch=%d parameter sweep per GPU, fp64 (BASELINE.json configs[3]: the "
                "whole 8-GPU sweep of 32768 on one GPU by default)"),
    # configs[4] is stated as batch=4096 over 8 GPUs: the largest single-GPU configuration is the whole 4096 on one GPU
    # (16 instances per CU queued behind each other: the workgroup dispatcher hands a finished instance's CU to the next
    # one, so the 367-iteration straggler no longer IS the batch time); `--batch 512` = one GPU's share of the 8-GPU run
    "quadcopter": ("tenscalc_amd.problems.mpc_quadcopter", "mpc_quadcopter", "quadcopter_data", 4096, "ldsc4w", 512,
                   "mpc_quadcopter T=100 (1406x1406 KKT, 67 k instructions), batch=%d per GPU, fp64 (BASELINE.json "
                   "configs[4]: batch=4096; 512 = the per-GPU share over 8 GPUs)"),
}


def default_variant(config, var0, B):
    """the back-end a config runs on when --variant is not given, by this rank's batch.  l1l2: the tile machine needs the
    whole machine's worth of tiles -- its batch time is one tile's solve, 67-69 ms, up to 4 096 instances -- while the LDS walk
    queues instances of 2.8 ms on 256 CUs: 2 048 / 4 096 / 8 192 per GPU: 24.1 / 47.1 / 91.9 ms `ldsc4` against 66.6 / 68.5 /
    78.8 ms `interp8` (profiles/r06_l1l2_share.log), so one GPU's share of configs[2] over 8 GPUs (2 048) runs on `ldsc4`.
    mpc_dcmotor and mpc_quadcopter keep their variant at every share."""
    if config == "l1l2" and B <= 6144:
        return "ldsc4"
    return var0


def bench_config(args, R):
    """python bench.py --config l1l2|dcmotor|quadcopter [--variant V] [--batch B | --total T]: one step = one batched
    solve of this rank's instances with the inputs resident in HBM; same timing contract and JSON fields as the sls line.
    Every instance of the batch is a distinct draw of the config's data generator (seed = its global instance id), so
    the iteration-count spread -- what makes a tile or a launch wait for its slowest instance -- is the real one.
    Returns rank 0's JSON object."""
    import importlib
    import numpy as np
    import torch
    import __graft_entry__ as g
    from tenscalc_amd.runtime import HipSolver
    from tenscalc_amd.sharding import pack_results, reduce_timing, shard_range
    rank, world, local, dev, dist = R.rank, R.world, R.local, R.dev, R.dist
    # TC_BENCH_STANDIN: the same control flow -- shards of a --total, padding of unequal shards, the one gather, the
    # reduction -- with a stand-in for the per-rank solver and gloo on CPU tensors (tests/test_bench_launcher.py)
    standin = R.standin
    modname, mk, datafn, B0, var0, ncpu, label = CONFIGS[args.config]
    _t = [time.perf_counter()]

    def phase(name):         # TC_BENCH_TIMING=1: where the wall time of a config sub-line goes (stderr)
        if os.environ.get("TC_BENCH_TIMING") and rank == 0:
            now = time.perf_counter()
            print("[bench %s] %-28s %7.2f s" % (args.config, name, now - _t[0]), file=sys.stderr, flush=True)
            _t[0] = now
    mod = importlib.import_module(modname)
    SOLVE = mod.SOLVE_ARGS
    if args.total:
        lo, hi = shard_range(args.total, rank, world)
        B, Bmax = hi - lo, -(-args.total // world)
    else:
        B = Bmax = args.batch or B0
        lo = rank * B
    variant = args.variant or default_variant(args.config, var0, Bmax)      # any key of __graft_entry__.VARIANT_KW
    # (mpc_dcmotor: the compiled LDS walk with four instances resident per CU is the fastest bit-exact back-end at both the
    # whole sweep on one GPU -- 21.2 ms against 24.7 `strict`, 21.8 `interp4` -- and one GPU's share of it -- 3.1 ms per 4096
    # against 5.3 `w32x`: profiles/r05_dcmotor_variants.log)
    spec = getattr(mod, mk)(folder=g.GENERATED)
    kw = dict(g.VARIANT_KW[variant])
    if not standin:
        if rank == 0:
            spec.build(**kw)
        R.barrier()
        so = spec.build(**kw)
    phase("tape + class library")
    names = [n for n, _ in spec.parameters + spec.variables]
    ds = [getattr(mod, datafn)(lo + k) for k in range(B)]                     # distinct instances, seed = global id
    stacked = {n: np.stack([np.asarray(d[n], dtype=float).reshape(-1, order="F") for d in ds]) for n in names}
    del ds
    P = max(1, int(args.pipeline))
    objs = [(_StandInConfigSolver(B, spec) if standin else HipSolver(so, B=B, device=local)) for _ in range(P)]
    for o_ in objs:
        for n in names:
            o_.set(n, stacked[n])
    obj = objs[0]
    phase("data, handles, inputs")

    class _NoStream:        # stand-in: nothing is asynchronous
        cuda_stream = None

        def wait_stream(self, other):
            pass
    # one torch stream per slot (P == 1: torch's current stream, which is where the kernel events are recorded)
    if standin:
        streams = [_NoStream() for _ in range(P)]
        cur_stream = _NoStream
    else:
        streams = [torch.cuda.current_stream()] if P == 1 else [torch.cuda.Stream(device=dev) for _ in range(P)]
        cur_stream = torch.cuda.current_stream
    ae = SOLVE.get("addEye2Hessian", (1e-9, 1e-9))
    nout = sum(o["numel"] for o in obj.outputs)
    views = []
    for o_ in objs:
        if standin:
            views.append((o_.O, o_.ST, o_.IT))
            continue
        optr, sptr, iptr = o_.device_views()
        views.append((torch.as_tensor(_DevArray(optr, (nout, B), "<f8"), device=dev),
                      torch.as_tensor(_DevArray(sptr, (B,), "<i4"), device=dev),
                      torch.as_tensor(_DevArray(iptr, (B,), "<i4"), device=dev)))
    O, ST, IT = views[(args.steps - 1) % P]           # the handle of the last step

    def packed():
        p_ = pack_results(O, ST, IT)
        if B < Bmax:
            pad = torch.zeros((nout + 2, Bmax - B), dtype=p_.dtype, device=dev)
            pad[nout] = -1.0
            p_ = torch.cat([p_, pad], 1)
        return p_
    for k in range(args.warmup * P):
        objs[k % P].solve_async(SOLVE["mu0"], SOLVE["maxIter"], ae, stream=streams[k % P].cuda_stream)
    evs = [] if standin else [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                              for _ in range(args.steps)]
    gathered = None
    if world > 1:      # one untimed gather of the same shape: RCCL sets up its channels at the first collective
        gathered = torch.empty((world, nout + 2, Bmax), dtype=torch.float64, device=dev)
        for s_ in streams:
            cur_stream().wait_stream(s_)
        dist.all_gather_into_tensor(gathered.view(world * (nout + 2), Bmax), packed().contiguous())
    R.barrier()
    R.sync()
    t0 = time.perf_counter()
    for k in range(args.steps):
        s_ = streams[k % P]
        if evs:
            evs[k][0].record(s_)
        objs[k % P].solve_async(SOLVE["mu0"], SOLVE["maxIter"], ae, stream=s_.cuda_stream)
        if evs:
            evs[k][1].record(s_)
    if P > 1:
        for s_ in streams:
            cur_stream().wait_stream(s_)
    if world > 1:      # the path's only exchange: one gather of status / iteration counts / outputs after the K steps
        dist.all_gather_into_tensor(gathered.view(world * (nout + 2), Bmax), packed().contiguous())
    R.sync()
    if world > 1:
        R.barrier()
        R.sync()
    elapsed = time.perf_counter() - t0
    ok_all = [int((v[1] == 0).sum().item()) for v in views[:min(P, args.steps)]]      # status of EVERY handle in flight
    elapsed, ok = reduce_timing(elapsed, float(np.mean(ok_all)), world, dev)
    phase("warm-up + timed steps")
    out = None
    calib = None
    if rank == 0 and not standin and variant.startswith("lds") and P == 1 and not getattr(args, "no_calib", False):
        # the time of ONE Newton iteration of ONE instance on this back-end, measured: the first <= 256 instances, one per
        # CU, in a batch of their own -- its time is its slowest instance's (see iteration_spread)
        Bc = min(B, 256)
        oc = HipSolver(so, B=Bc, device=local)
        for n in names:
            oc.set(n, stacked[n][:Bc])
        oc.solve(**SOLVE)
        _, itc, tmc = oc.solve(**SOLVE)
        calib = (float(tmc), int(np.asarray(itc).max()), Bc, max(1, int(160 * 1024 // (8 * int(kw.get("lds_slots", 20000))))))
        oc.close()
        phase("per-iteration calibration")
    if rank == 0:
        it = IT.cpu().numpy()
        st = ST.cpu().numpy()
        nin = sum(int(np.prod(s_)) if len(s_) else 1 for _, s_ in spec.parameters + spec.variables)
        algo = 8 * (nin + nout) + 8                               # inputs + outputs + (status, iter)
        if P == 1 and evs:
            kavg = float(np.mean([a.elapsed_time(b_) for a, b_ in evs])) * 1e-3
            kernel_note = "HIP events around each launch on the launch stream"
        else:       # spans of overlapping launches include the time spent sharing the chip: use the whole-region rate
            kavg = elapsed / args.steps
            kernel_note = "%d batches in flight: elapsed / steps (per-launch event spans overlap and are not reported)" % P
        achieved = algo * B / kavg / 1e9
        total = args.total or B * world
        out = {"metric": "IPM solves/sec (whole node), %s batch" % args.config, "value": ok * args.steps / elapsed,
               "unit": "solves/s", "n_gpus": len(set(R.devices)), "ranks": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
               "scaling": "strong" if args.total else "weak", "vs_baseline": None,
               "dtype": "f64", "data": "synthetic",
               "config": {"workload": (label % B) + (" [--total %d sharded over %d GPU(s)]" % (args.total, world) if args.total else ""),
                          "batch_per_gpu": B, "batch_total": total, "mu0": SOLVE["mu0"], "maxIter": SOLVE["maxIter"],
                          "variant": variant, **parity_fields(g.PARITY_CLASS.get(variant, "solver-tolerance"), spec.tape, spec.defines),
                          "pipeline": P, "distinct_instances": B,
                          "solved_fraction": ok / total, "iter_median": float(np.median(it)),
                          "iter_max": int(it.max()), **iteration_spread(it, variant, B, kavg, calib),
                          "devices": R.devices,
                          "sharding": "independent instances per rank; one RCCL all_gather of the results after the K steps"},
               "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                            "frac": achieved / HBM_PEAK_GBPS, "traffic": measured_traffic(args.config + "_" + variant, B),
                            "traffic_from": traffic_from(args.config + "_" + variant, B),
                            "kernel": "tc_solve_kernel", "kernel_ms_avg": kavg * 1e3, "kernel_ms_from": kernel_note,
                            "algorithmic_bytes_per_launch": algo * B, "algorithmic_bytes_per_solve": algo,
                            # SURVEY.md 8d's formula, counted from the problem statement: mpc_dcmotor has 42 + 89 input and
                            # 241 output doubles (mpc_dcmotor.m:118-126: J, Jx2, Ju2, t[30], u[29], x[60], ref[30], xWarm[60],
                            # uWarm[29]) = 2 984 B; the survey's 3 224 counts 271 outputs, 30 more than the file declares
                            "algorithmic_doubles": {"inputs": nin, "outputs": nout, "status_iter": 1}}}
        if standin:
            out["data"] = "STAND-IN: no solver ran (TC_BENCH_STANDIN rehearsal of the multi-rank control flow)"
            out["standin"] = True
            if gathered is not None:       # what the one gather delivered to rank 0, for the rehearsal's checks
                out["gathered"] = {"status": gathered[:, nout].to(torch.int32).tolist(),
                                   "checksum": gathered[:, 0].tolist(), "shard": [lo, lo + B]}
        if not args.no_cpu and world == 1 and not standin:
            out["cpu_baseline"] = cpu_baseline_config(spec, args.config, stacked, names, min(B, ncpu), SOLVE, st, it,
                                                      budget_s=getattr(args, "cpu_budget_s", 8.0))
    phase("cpu baseline + line")
    for o_ in objs:
        o_.close()
    phase("close")
    return out


def _cpu_config_worker(rank, parts, sub, n, mu0, maxIter, budget_s, q):
    """one host core of the all-cores baseline of a config: the generated-C twin over this worker's instances, repeated
    until the budget is spent (a pass cannot be interrupted)"""
    import numpy as np  # noqa: F401
    from oracle.c_emitter import CpuTwin
    tw = CpuTwin.from_parts(*parts)
    tw.solve_batch({k: v[:1] for k, v in sub.items()}, 1, mu0=mu0, maxIter=maxIter)       # load, page in
    t0 = time.perf_counter()
    solved = 0
    while True:
        _, st, _ = tw.solve_batch(sub, n, mu0=mu0, maxIter=maxIter)
        solved += int((st == 0).sum())
        if time.perf_counter() - t0 >= budget_s:
            break
    q.put((rank, time.perf_counter() - t0, solved))


def cpu_all_cores_config(twin, sub, n, SOLVE, per_pass_s, budget_s=4.0):
    """the CPU path of a config on every host core of the box's share (one worker process per core, the generated C keeps
    its state in globals): worker r solves the r-th slice of the sample, sized so that a pass takes about a second,
    again and again for `budget_s`; aggregate = solves of all workers / slowest worker's time.  MEASURED, not the one-core
    figure times the core count."""
    import multiprocessing as mp
    P = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    P = max(1, min(P, int(os.environ.get("TC_CPU_WORKERS", "16"))))
    per = max(1, min(n // P if n >= P else 1, int(max(1.0, 1.0 / max(per_pass_s / max(n, 1), 1e-9)))))      # ~1 s per pass
    ps, res = [], []
    try:
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        for r in range(P):
            lo = (r * per) % max(1, n - per + 1)
            ps.append(ctx.Process(target=_cpu_config_worker,
                                  args=(r, twin.parts(), {k: v[lo:lo + per].copy() for k, v in sub.items()}, per,
                                        SOLVE["mu0"], SOLVE["maxIter"], budget_s, q)))
        for p_ in ps:
            p_.start()
        deadline = time.monotonic() + 120 + 4 * budget_s
        while len(res) < P:
            try:
                res.append(q.get(timeout=1.0))
            except Exception:
                if any(p_.exitcode not in (None, 0) for p_ in ps) or time.monotonic() > deadline:
                    return None
    except Exception:
        return None
    finally:
        for p_ in ps:
            if p_.is_alive() and len(res) < P:
                p_.terminate()
        for p_ in ps:
            p_.join(timeout=10)
    wall = max(r[1] for r in res)
    return {"value_all_cores": sum(r[2] for r in res) / wall, "cores_all": P,
            "sample_all_cores": "%d worker processes, each its own %d instances of the sample over and over for %.1f s "
                                "(slowest worker %.2f s)" % (P, per, budget_s, wall)}


def cpu_baseline_config(spec, which, stacked, names, n, SOLVE, st, it, budget_s=8.0):
    """bounded sample of the same workload on one host core: the generated-C twin (the tape as reference-style C + the
    reference's ipmPD_CSsolver.c, gcc -O1; for the big tapes every value in the scratchbook and the group bodies cut
    into translation units compiled side by side, oracle/c_emitter.py), else the oracle's tape interpreter"""
    import numpy as np
    sub = {k: v[:n] for k, v in stacked.items()}
    twin = None
    try:        # prebuilt in the development container for every config (oracle/_ref/lib<cls>_cpu.so)
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
    if n > 16:          # pilot: size the sample so that one pass fits the budget (a pass cannot be interrupted)
        n0 = 8
        sub0, sub_all = {k: v[:n0] for k, v in sub.items()}, sub
        sub = sub0
        t0 = time.perf_counter()
        nn, n = n, n0
        run()
        rate = n0 / max(time.perf_counter() - t0, 1e-6)
        n = int(min(nn, max(n0, rate * budget_s)))
        sub = {k: v[:n] for k, v in sub_all.items()}
    t0 = time.perf_counter()
    reps = 0
    while reps == 0 or (time.perf_counter() - t0 < budget_s and reps < 64):        # ~10 s of CPU work
        out, st2, it2 = run()
        reps += 1
    dt = (time.perf_counter() - t0) / reps
    res = {"value": float((st2 == 0).sum()) / dt, "unit": "solves/s", "cores": 1, "kind": "port",
           "sample": "first %d instances of the benchmark batch, %s, sequential on 1 of %d host cores, %.2f s per pass"
                     % (n, how, os.cpu_count(), dt),
           "same_status_iter": bool(np.array_equal(st[:n], st2) and np.array_equal(it[:n], it2))}
    if twin is not None and not os.environ.get("TC_BENCH_NO_ALLCORES"):
        allc = cpu_all_cores_config(twin, sub, n, SOLVE, dt, budget_s=min(4.0, budget_s))
        if allc:
            res.update(allc)
    if which == "l1l2" and twin is not None:
        # the GPU class orders the KKT factorisation for a short dependency chain (1.2x the multiply-adds); a CPU would
        # keep the fill-minimising order: the same sample through the twin of THAT tape, reported beside the first
        try:
            import importlib
            from oracle.c_emitter import CpuTwin
            import __graft_entry__ as g
            tw2 = CpuTwin(importlib.import_module("tenscalc_amd.problems.l1l2").l1l2(
                folder=g.GENERATED, ldlOrder="mindeg", classname="tmpC_minl1l2_mindeg"))
            t0 = time.perf_counter()
            _, st3, _ = tw2.solve_batch(sub, n, mu0=SOLVE["mu0"], maxIter=SOLVE["maxIter"])
            res["value_mindeg_order"] = float((st3 == 0).sum()) / (time.perf_counter() - t0)
        except (FileNotFoundError, OSError):
            pass
    return res


if __name__ == "__main__":
    main()
