"""GPU box (one MI355X): the RCCL branch of bench.py with a single rank.  torch.distributed.run starts ONE rank,
TC_BENCH_FORCE_DIST makes it form the process group anyway (backend "nccl" = RCCL, device_id given), so init,
barrier(device_ids), the asynchronous all_gather_into_tensor of the result exchange, all_reduce of time / solved count and
the teardown all execute on real hardware -- the code an 8-GPU run goes through, minus the second device."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("extra", [[], ["--gather-every-step"], ["--total", "4096"]])
def test_bench_single_rank_over_rccl(extra):
    env = dict(os.environ, TC_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "1", "--no-cpu",
           "--no-configs", "--sustain", "0"] + (extra if "--total" in extra else ["--batch", "4096"] + extra)
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 1 and len(d["config"]["devices"]) == 1
    assert d["config"]["gathered_solved_last_step"] == 4096 and d["config"]["solved_fraction"] == 1.0
    assert d["scaling"] == ("strong" if "--total" in extra else "weak")
