"""GPU: the C-ABI bound from plain C (tests/cabi_harness.c: gcc, dlopen, include/tenscalc_hip.h; no
Python or torch in that process) reproduces the golden vectors bit for bit through the strict class
library, and its reference-named single-instance symbols agree with the batched path."""
import os
import subprocess

import numpy as np
import pytest

from tenscalc_amd.problems.sls import sls_batch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def test_c_host_reproduces_golden_vectors(tmp_path):
    import __graft_entry__ as g
    spec = g.sls_spec("strict")
    so = spec.build(**g.VARIANT_KW["strict"])
    exe = str(tmp_path / "cabi_harness")
    subprocess.run(["gcc", "-O1", "-std=gnu99", "-I", os.path.join(ROOT, "include"), os.path.join(HERE, "cabi_harness.c"),
                    "-o", exe, "-ldl"], check=True)
    gold = np.load(os.path.join(HERE, "golden", "sls_N100_n8.npz"))
    B = len(gold["seeds"])
    A, b, x0 = sls_batch(B, seed0=int(gold["seeds"][0]))
    inp, out = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(inp, "wb") as f:
        f.write(np.int64(B).tobytes())
        f.write(np.ascontiguousarray(np.transpose(A, (0, 2, 1))).tobytes())     # [B][n][N]: MATLAB [N,n] per instance
        f.write(np.ascontiguousarray(b).tobytes())
        f.write(np.ascontiguousarray(x0).tobytes())
    r = subprocess.run([exe, so, spec.classname, inp, out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    raw = open(out, "rb").read()
    st = np.frombuffer(raw, dtype=np.int32, count=B)
    it = np.frombuffer(raw, dtype=np.int32, count=B, offset=4 * B)
    J = np.frombuffer(raw, dtype=np.float64, count=B, offset=8 * B)
    x = np.frombuffer(raw, dtype=np.float64, count=8 * B, offset=16 * B).reshape(B, 8)
    assert np.array_equal(st, gold["status"]) and np.array_equal(it, gold["iter"])
    assert np.array_equal(J, gold["J"]) and np.array_equal(x, gold["x"])
