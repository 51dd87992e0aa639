"""The parity class bench.py / smoke() print is a claim about the HOST's libm: the device restates glibc 2.35's pow
(csrc/tc_glibcpow.hip.inc) for `pow(mu,.5)` (ipmPD_CSsolver.c:801) and the I_Dsqrt / I_DDsqrt opcodes
(instructionsTableUTHash.c:1418,1424).  tools/powcheck.py checks it against live calls of pow where the label is printed;
on a mismatch the label becomes "solver-tolerance (host libm differs)"."""
import json
import os
import subprocess
import sys

import numpy as np

from tenscalc_amd import tcalculus as tc
from tenscalc_amd.tools import powcheck

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_build_matches_this_hosts_pow():
    r = powcheck.pow_parity(n=100_000, device=False)
    assert r["ok"] and r["n"] == 100_000 and r["libc"].startswith("glibc")


def _specs():
    from tenscalc_amd.problems.flops import flops_problem
    from tenscalc_amd.problems.mpc_quadcopter import mpc_quadcopter
    from tenscalc_amd.problems.sls import sls
    tc.clear()
    return flops_problem(6), sls(N=12, n=3, classname="tmp_lbl_sls"), mpc_quadcopter(T=4, classname="tmp_lbl_quad")


def test_which_classes_depend_on_pow():
    free, box, quad = _specs()
    assert powcheck.tape_calls_pow(free.tape, free.defines) == (False, False)       # nF = 0: no mu, no pow
    assert powcheck.tape_calls_pow(box.tape, box.defines) == (False, True)          # mu update: pow(mu,.5)
    assert powcheck.tape_calls_pow(quad.tape, quad.defines) == (True, True)         # norms: .5*pow(x,-.5), -.25*pow(x,-1.5)


def test_a_faked_mismatch_downgrades_the_label(monkeypatch):
    free, box, quad = _specs()
    good = powcheck.pow_parity(n=20_000, device=False)
    for spec in (free, box, quad):
        assert powcheck.checked_parity_class("bit-exact", spec.tape, spec.defines, check=good)[0] == "bit-exact"
        assert powcheck.checked_parity_class("solver-tolerance", spec.tape, spec.defines, check=good) == ("solver-tolerance", None)
    # a host whose libm rounds pow differently on one argument in a thousand (what glibc <= 2.27 vs 2.35 looks like)
    real = powcheck.host_pow

    def other_libm(x):
        a, b, h = real(x)
        k = np.arange(x.size) % 1000 == 7
        return np.where(k, np.nextafter(a, np.inf), a), b, np.where(k, np.nextafter(h, 0.0), h)
    monkeypatch.setattr(powcheck, "host_pow", other_libm)
    bad = powcheck.pow_parity(n=20_000, device=False)
    assert not bad["ok"] and bad["neg_half"] == 20 and bad["half"] == 20 and bad["neg_three_halves"] == 0
    assert powcheck.checked_parity_class("bit-exact", free.tape, free.defines, check=bad)[0] == "bit-exact"      # calls no pow
    assert powcheck.checked_parity_class("bit-exact", box.tape, box.defines, check=bad)[0] == powcheck.DOWNGRADED
    assert powcheck.checked_parity_class("bit-exact", quad.tape, quad.defines, check=bad)[0] == powcheck.DOWNGRADED
    only_neg = dict(bad, half=0)
    assert powcheck.checked_parity_class("bit-exact", box.tape, box.defines, check=only_neg)[0] == "bit-exact"
    assert powcheck.checked_parity_class("bit-exact", quad.tape, quad.defines, check=only_neg)[0] == powcheck.DOWNGRADED


def test_classes_that_call_exp_log_sin_cos_atan_are_checked_too(monkeypatch):
    """round 6: the other libm functions of the ISA are restated for the device as well; a class that evaluates them keeps the
    label only if the (host build of the) restatement returns this host's bits, and a faked mismatch downgrades it"""
    from tenscalc_amd.problems.small import opcode_zoo
    tc.clear()
    zoo = opcode_zoo(folder="/tmp/lbl_zoo")
    free, box, quad = _specs()
    assert powcheck.tape_calls_libm(zoo.tape) == ["atan", "cos", "exp", "log", "sin"]
    assert powcheck.tape_calls_libm(box.tape) == [] and powcheck.tape_calls_libm(quad.tape) == []
    lm = powcheck.libm_parity(powcheck.LIBM_FUNCTIONS, n=50_000, device=False)
    assert lm == {k: 0 for k in powcheck.LIBM_FUNCTIONS}
    good = powcheck.pow_parity(n=20_000, device=False)
    monkeypatch.setattr(powcheck, "libm_parity", lambda names, **kw: {k: 0 for k in names})
    label, chk = powcheck.checked_parity_class("bit-exact", zoo.tape, zoo.defines, check=good)
    assert label == "bit-exact" and set(chk["libm"]) == set(powcheck.LIBM_FUNCTIONS)
    monkeypatch.setattr(powcheck, "libm_parity", lambda names, **kw: {k: (3 if k == "cos" else 0) for k in names})
    assert powcheck.checked_parity_class("bit-exact", zoo.tape, zoo.defines, check=good)[0] == powcheck.DOWNGRADED
    assert powcheck.checked_parity_class("bit-exact", box.tape, box.defines, check=good)[0] == "bit-exact"      # calls none of them


def test_bench_line_carries_the_check():
    """the stand-in rehearsal of bench.py (no GPU) prints the checked label and the outcome of the check"""
    env = dict(os.environ, TC_BENCH_STANDIN="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "64", "--no-cpu"],
                       capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    cfg = line["config"]
    assert cfg["parity_class"] == "bit-exact" and cfg["libm_check"]["ok"] and cfg["libm_check"]["n"] >= 100_000
