"""The MATLAB gateway integration/tc_mex.c, compiled against a stand-in for the mex API (tests/mock_mex/) and driven
the way MATLAB drives a mex file: argument validation + handle life cycle on the CPU (stale, made-up and destroyed
handles are error messages, a failing create does not leak its slot), and on the GPU a create / set / solve / get /
destroy round trip through the strict sls class library that reproduces the golden vectors bit for bit."""
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _build(tmp_path):
    exe = str(tmp_path / "mex_driver")
    subprocess.run(["gcc", "-O1", "-std=gnu99", "-Wall", "-Werror", "-I", os.path.join(HERE, "mock_mex"),
                    "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "integration", "tc_mex.c"),
                    os.path.join(HERE, "mock_mex", "mex_driver.c"), "-o", exe, "-ldl"], check=True)
    return exe


def test_gateway_argument_validation_and_handle_life_cycle(tmp_path):
    from oracle import build as obuild
    notalib = obuild.build_oracle()          # a real shared library that has no tc_* entry points
    r = subprocess.run([_build(tmp_path), "cpu", notalib], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "MEX_CPU_OK" in r.stdout, (r.returncode, r.stdout, r.stderr)


@pytest.mark.gpu
def test_gateway_round_trip_reproduces_golden_vectors(tmp_path):
    import __graft_entry__ as g
    from tenscalc_amd.problems.sls import sls_batch
    spec = g.sls_spec("strict")
    so = spec.build(**g.VARIANT_KW["strict"])
    gold = np.load(os.path.join(HERE, "golden", "sls_N100_n8.npz"))
    B = len(gold["seeds"])
    A, b, x0 = sls_batch(B, seed0=int(gold["seeds"][0]))
    inp, out = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(inp, "wb") as f:
        f.write(np.int64(B).tobytes())
        f.write(np.ascontiguousarray(np.transpose(A, (0, 2, 1))).tobytes())     # [B][n][N] = MATLAB [N,n,B]
        f.write(np.ascontiguousarray(b).tobytes())
        f.write(np.ascontiguousarray(x0).tobytes())
    r = subprocess.run([_build(tmp_path), "gpu", so, inp, out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "MEX_GPU_OK" in r.stdout, (r.returncode, r.stdout, r.stderr)
    raw = open(out, "rb").read()
    st = np.frombuffer(raw, dtype=np.int32, count=B)
    it = np.frombuffer(raw, dtype=np.int32, count=B, offset=4 * B)
    J = np.frombuffer(raw, dtype=np.float64, count=B, offset=8 * B)
    x = np.frombuffer(raw, dtype=np.float64, count=8 * B, offset=16 * B).reshape(B, 8)
    assert np.array_equal(st, gold["status"]) and np.array_equal(it, gold["iter"])
    assert np.array_equal(J, gold["J"]) and np.array_equal(x, gold["x"])
