"""Tuning aid (GPU box): time one sls class built with ad-hoc emitter options.
   python scripts/try_variant.py <base variant> key=value,... [ENV=value ...]"""
import json
import os
import subprocess
import sys

base, opts = sys.argv[1], sys.argv[2]
env = dict(os.environ)
for kv in sys.argv[3:]:
    k, v = kv.split("=")
    env[k] = v
env["TC_EXPERIMENT"] = opts
r = subprocess.run([sys.executable, "bench.py", "--no-cpu", "--variant", base, "--steps", "10"], env=env, capture_output=True, text=True)
try:
    d = json.loads(r.stdout.strip().splitlines()[-1])
    print(base, opts, " ".join(sys.argv[3:]), "->", round(d["ms_per_step"], 4), "ms, solved", d["config"]["solved_fraction"], flush=True)
except Exception:
    print(base, opts, "FAILED", r.stderr[-600:], flush=True)
