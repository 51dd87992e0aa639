"""bench.py --gpus N end to end where there is no GPU: the parent starts N fresh rank processes through
torch.distributed.run, the ranks form a gloo process group, shard the batch (weak: --batch per rank; strong: --total
through sharding.shard_range), run the K steps with a STAND-IN for the per-rank solver (TC_BENCH_STANDIN: not a solver,
not the oracle), exchange results once, and rank 0's JSON line comes back through the parent.  What is checked is what
the driver relies on: n_gpus from the process group, distinct devices, summed solves, max-over-ranks time."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*argv, env_extra=None, timeout=300):
    env = dict(os.environ, TC_BENCH_STANDIN="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + [str(a) for a in argv],
                       capture_output=True, text=True, env=env, timeout=timeout, cwd=ROOT)
    return r


def test_gpus_2_weak_scaling_line(native_built):
    r = _bench("--gpus", 2, "--steps", 3, "--warmup", 1, "--batch", 64, "--no-cpu")
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                  # ONE JSON line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["standin"] is True and "STAND-IN" in d["data"]
    assert len(set(d["config"]["devices"])) == 2
    assert d["config"]["batch_per_gpu"] == 64 and d["config"]["batch_total"] == 128
    assert d["config"]["solved_fraction"] == 1.0
    assert d["config"]["gathered_solved_last_step"] == 128          # both ranks' results arrived on rank 0
    # value = solved instances of ALL ranks * steps / max-over-ranks time
    assert abs(d["value"] - 128 * 3 / (d["ms_per_step"] * 3e-3)) <= 1e-6 * d["value"]


def test_gpus_3_strong_scaling_unequal_shards(native_built):
    r = _bench("--gpus", 3, "--steps", 2, "--warmup", 1, "--total", 100, "--no-cpu")
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 3 and d["scaling"] == "strong"
    assert d["config"]["batch_total"] == 100 and d["config"]["batch_per_gpu"] == 34       # rank 0 of 34 / 33 / 33
    assert d["config"]["gathered_solved_last_step"] == 100                                  # padding columns not counted
    assert d["config"]["solved_fraction"] == 1.0


def test_world_size_must_match_gpus(native_built):
    r = _bench("--gpus", 2, "--steps", 1, "--warmup", 0, "--batch", 8, "--no-cpu",
               env_extra=dict(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr


def test_default_is_one_gpu_and_batch_total_exclusive(native_built):
    r = _bench("--batch", 8, "--total", 8)
    assert r.returncode != 0 and "exclude" in r.stderr


def test_config_runs_are_rehearsed_too(native_built):
    """`bench.py --gpus N --config dcmotor --total T` -- BASELINE configs[3] / [4] are stated as 8-GPU totals -- through the
    same stand-in: contiguous unequal shards of distinct instances (seed = global id), padding columns carry status -1,
    ONE gather delivers every rank's results to rank 0, solves summed over the ranks, max-over-ranks time"""
    r = _bench("--gpus", 3, "--config", "dcmotor", "--total", 50, "--steps", 2, "--warmup", 1, "--no-cpu", timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["standin"] is True and d["n_gpus"] == 3 and d["scaling"] == "strong" and d["steps"] == 2
    cfg = d["config"]
    assert cfg["batch_total"] == 50 and cfg["batch_per_gpu"] == 17 and cfg["solved_fraction"] == 1.0      # 17 / 17 / 16
    assert "--total 50 sharded over 3" in cfg["workload"] and len(set(cfg["devices"])) == 3
    g_ = d["gathered"]
    st = g_["status"]
    assert [len(row) for row in st] == [17, 17, 17] and g_["shard"] == [0, 17]
    assert all(v == 0 for v in st[0] + st[1] + st[2][:16]) and st[2][16] == -1               # the one padding column
    sums = g_["checksum"]
    flat = sums[0] + sums[1] + sums[2][:16]
    assert len(set(round(v, 9) for v in flat)) == 50                                             # 50 distinct instances arrived
    assert abs(d["value"] - 50 * 2 / (d["ms_per_step"] * 2e-3)) <= 1e-6 * d["value"]


def test_config_single_rank_standin(native_built):
    r = _bench("--config", "dcmotor", "--batch", 8, "--steps", 1, "--warmup", 0, "--no-cpu", timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["standin"] is True and d["n_gpus"] == 1 and d["config"]["batch_total"] == 8 and "gathered" not in d


def test_iteration_spread_and_traffic_provenance():
    """the fields that make a batch's tail readable (bench.py: iteration_spread) and say where replayed counters come from
    (traffic_from): floors from a measured per-iteration time, residents per CU, lockstep waste of a tile back-end"""
    import numpy as np
    sys.path.insert(0, ROOT)
    import bench
    rs = np.random.RandomState(0)
    it = np.r_[rs.poisson(28, 4095), 765]
    s = bench.iteration_spread(it, "ldsc4w", 4096, 0.400, (0.050, 180, 256, 1))
    t_iter = 0.050 / 180
    assert abs(s["ms_per_iteration_one_instance"] - t_iter * 1e3) < 1e-12 and s["residents_per_cu"] == 1
    assert abs(s["straggler_floor_ms"] - 765 * t_iter * 1e3) < 1e-9
    assert abs(s["throughput_floor_ms"] - it.sum() * t_iter * 1e3 / 256) < 1e-9
    assert s["straggler_position"] == 1.0 and abs(s["batch_over_floor"] - 400.0 / s["straggler_floor_ms"]) < 1e-9
    assert sum(s["iter_hist"]["counts"]) == it.size and s["iter_total"] == int(it.sum())
    s4 = bench.iteration_spread(it, "ldscs", 4096, 0.400, (0.050, 180, 256, 4))
    assert abs(s4["throughput_floor_ms"] * 4 - s["throughput_floor_ms"]) < 1e-9           # four workgroups per CU
    assert "straggler_floor_ms" not in bench.iteration_spread(it, "ldsc4w", 4096, 0.400, None)      # --no-calib: no floors claimed
    tile = bench.iteration_spread(rs.poisson(29, 16384), "interp8", 16384, 0.130, None)
    assert 0 < tile["lockstep_waste"] < 0.5 and tile["tile_iter_max_mean"] > tile["iter_mean"]
    tf = bench.traffic_from("w8x", 65536)
    assert tf["file"] == "profiles/traffic.json" and tf["entry"] == "w8x_B65536" and "NOT a measurement of this run" in tf["how"]
    assert bench.traffic_from("no_such_variant", 1) is None and bench.measured_traffic("no_such_variant", 1) is None


def test_default_variant_by_share():
    """--config runs without --variant pick the back-end by THIS rank's batch: one GPU's share of configs[2] over 8 GPUs (2 048
    instances of l1l2) runs on the LDS walk, the whole 16 384 on the tile machine (profiles/r06_l1l2_share.log); the MPC configs
    keep their variant; every variant named is one build() provides for that config"""
    sys.path.insert(0, ROOT)
    import bench
    import __graft_entry__ as g
    var0 = {k: v[4] for k, v in bench.CONFIGS.items()}
    assert bench.default_variant("l1l2", var0["l1l2"], 16384) == "interp8"
    assert bench.default_variant("l1l2", var0["l1l2"], 2048) == "ldsc4"
    assert bench.default_variant("l1l2", var0["l1l2"], 8192) == "interp8"
    for cfg in ("dcmotor", "quadcopter"):
        assert {bench.default_variant(cfg, var0[cfg], b) for b in (512, 4096, 32768)} == {var0[cfg]}
    src = open(os.path.join(ROOT, "__graft_entry__.py")).read()
    for tag in ("l1l2_ldsc4", "l1l2_interp8", "dcmotor_" + var0["dcmotor"], "quadcopter_" + var0["quadcopter"]):
        assert tag in src or tag.split("_", 1)[1] in g.VARIANT_KW
