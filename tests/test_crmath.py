"""The correctly rounded x^(-1/2), x^(-3/2) of tenscalc_amd/csrc/tc_crmath.hip.inc (device replacement for the
reference's `.5*pow(x,-.5)` / `-.25*pow(x,-1.5)`, instructionsTableUTHash.c:1418,1424): the host build of the same
file against exact integer arithmetic, and against the host libm's pow."""
import ctypes
import math

import numpy as np
import pytest

from tenscalc_amd.tools.powcheck import build_dev, build_host, eval2 as _eval, sample      # noqa: E402


def exact_power(x, three):
    """RN(x^(-1/2)) (three=False) or RN(x^(-3/2)) for a positive finite double, by integer arithmetic."""
    m, e = math.frexp(x)
    m = int(m * (1 << 53))
    e -= 53                              # x = m * 2^e exactly
    if three:
        m, e = m ** 3, 3 * e
    K = 260 + (abs(e) // 2 + 2)          # y * 2^K = sqrt(2^(2K) / (m 2^e)) has > 200 significant bits
    if (2 * K - e) < 0:
        raise ValueError
    q = math.isqrt((1 << (2 * K - e)) // m)          # floor; the dropped fraction is far below bit 53
    nb = q.bit_length()
    sh = nb - 53
    hi, rest = q >> sh, q & ((1 << sh) - 1)
    half = 1 << (sh - 1)
    assert abs(rest - half) > 4, "too close to a midpoint for this check"
    if rest > half:
        hi += 1
    return math.ldexp(hi, sh - K)


def test_correctly_rounded_against_exact_arithmetic():
    L = build_host()
    x = sample(200_000, 1)
    a, b = _eval(L.crm_eval, x)
    ea = np.array([exact_power(float(v), False) for v in x])
    eb = np.array([exact_power(float(v), True) for v in x])
    assert np.array_equal(a, ea)
    assert np.array_equal(b, eb)


def test_edges_follow_pow():
    L = build_host()
    x = np.array([0.0, -0.0, -1.0, np.inf, np.nan, -np.inf, 4.0, 0.25, 2.0 ** -1074, 2.0 ** -1000, 2.0 ** -700,
                  2.0 ** 700, 2.0 ** 1000, 1.7976931348623157e308, 2.0 ** -1022, 1.0, 2.0 ** 680, 3e-320])
    a, b = _eval(L.crm_eval, x)
    pa, pb = _eval(L.libm_eval, x)
    with np.errstate(all="ignore"):
        for u, v in ((a, pa), (b, pb)):
            assert np.array_equal(np.isnan(u), np.isnan(v))
            ok = ~np.isnan(v)
            # subnormal results are rounded twice by ours: 1 ulp there
            assert np.all((u[ok] == v[ok]) | (np.abs(u[ok] - v[ok]) <= np.spacing(np.abs(v[ok])) * (np.abs(v[ok]) < 2.3e-308)))
    assert a[6] == 0.5 and b[6] == 0.125 and a[7] == 2.0 and b[7] == 8.0
    # extreme arguments against exact arithmetic
    for v in (2.0 ** -1000 * 1.37, 2.0 ** -700 * 1.9, 2.0 ** 700 * 1.1, 2.0 ** 1000 * 1.77, 3e-320, 1.7976931348623157e308):
        a1, b1 = _eval(L.crm_eval, np.array([v]))
        assert a1[0] == exact_power(v, False)
        if 2.0 ** -682 < v < 2.0 ** 681:
            assert b1[0] == exact_power(v, True)


def test_host_libm_agrees_except_where_it_misrounds():
    """glibc's pow is not correctly rounded; wherever it differs from ours, exact arithmetic must side with ours,
    and the differing fraction is small -- but not small enough for a solve that calls pow 1e4 times, which is why the
    device restates glibc's algorithm instead (csrc/tc_glibcpow.hip.inc)."""
    L = build_host()
    x = sample(2_000_000, 2)
    a, b = _eval(L.crm_eval, x)
    pa, pb = _eval(L.libm_eval, x)
    for ours, libm, three in ((a, pa, False), (b, pb, True)):
        d = np.nonzero(ours != libm)[0]
        assert d.size <= 5e-3 * x.size, d.size      # measured: 8e-4 (y=-.5), 4e-3 (y=-1.5)
        for i in d[:200]:
            assert ours[i] == exact_power(float(x[i]), three)
            assert abs(ours[i] - libm[i]) <= np.spacing(abs(ours[i]))


def test_host_build_of_the_glibc_restatement_is_the_hosts_pow():
    """csrc/tc_glibcpow.hip.inc compiled for the host returns the bits of THIS machine's pow(x,-.5), pow(x,-1.5):
    4e6 ordinary arguments, all binades, subnormal arguments, the special values."""
    L = build_host()
    r = np.random.RandomState(7)
    x = np.concatenate([sample(4_000_000, 4), 2.0 ** r.uniform(-1074, 1023, 200000), 2.0 ** r.uniform(-1074, -1020, 50000),
                        np.array([0.0, -0.0, -1.0, -3e-320, np.inf, -np.inf, np.nan, 1.0, 4.0, 5e-324, 2.2250738585072014e-308,
                                  1.7976931348623157e308, 2.0 ** -1000, 2.0 ** -683, 2.0 ** -682, 2.0 ** -340, 2.0 ** 700,
                                  2.0 ** 716, 2.0 ** 717])])
    a, b = _eval(L.glibc_emul_eval, x)
    pa, pb = _eval(L.libm_eval, x)
    assert np.array_equal(a, pa, equal_nan=True)
    assert np.array_equal(b, pb, equal_nan=True)
    # pow(mu,.5) of the Newton loops (ipmPD_CSsolver.c:801): NOT sqrt(mu) -- glibc differs from it on 8e-4 of arguments
    # (tc_glibc_pow_half: the hardware's sqrt unless x is within 1/16 ulp of a rounding boundary of sqrt, where glibc's
    # 0.52-ulp result may differ; else the restated algorithm)
    e, l = _eval(L.half_eval, x)
    assert np.array_equal(e, l, equal_nan=True)
    L.half_shortcuts.restype = ctypes.c_long
    k = L.half_shortcuts(np.ascontiguousarray(x).ctypes.data_as(ctypes.POINTER(ctypes.c_double)), ctypes.c_long(x.size))
    assert .8 < k / x.size < .9                     # 7/8 of the arguments take the shortcut
    with np.errstate(invalid="ignore"):
        assert 1e-4 < np.mean(l[x > 0] != np.sqrt(x[x > 0])) < 5e-3


@pytest.mark.gpu
def test_device_pow_is_the_hosts_pow_bit_for_bit():
    """the device evaluators' TC_POW_NEG (what I_Dsqrt / I_DDsqrt call) == pow() of the GPU box's own libm, and the
    device build of the correctly rounded pair == its host build"""
    import torch  # noqa: F401  (one HIP runtime per process)
    L = build_host()
    D = ctypes.CDLL(build_dev())
    r = np.random.RandomState(11)
    x = np.concatenate([sample(10_000_000, 12), 2.0 ** r.uniform(-1074, 1023, 500000),
                        np.array([0.0, -0.0, -1.0, np.inf, -np.inf, np.nan, 1.0, 4.0, 5e-324, 2.2250738585072014e-308,
                                  1.7976931348623157e308, 2.0 ** -1000, 2.0 ** -683, 2.0 ** 700])])
    dp = ctypes.POINTER(ctypes.c_double)
    o = [np.empty_like(x) for _ in range(4)]
    rc = D.crm_dev_eval(x.ctypes.data_as(dp), *[v.ctypes.data_as(dp) for v in o], ctypes.c_long(x.size))
    assert rc == 0
    pa, pb = _eval(L.libm_eval, x)
    ca, cb = _eval(L.crm_eval, x)
    n1, n2 = int((~((o[0] == pa) | (np.isnan(pa) & np.isnan(o[0])))).sum()), int((~((o[1] == pb) | (np.isnan(pb) & np.isnan(o[1])))).sum())
    print("\ndevice TC_POW_NEG vs this host's pow on %d arguments: %d / %d mismatches (y=-.5 / y=-1.5); "
          "correctly rounded pair vs pow: %.2e / %.2e of arguments differ" %
          (x.size, n1, n2, np.mean(o[2] != pa), np.mean(o[3] != pb)))
    assert n1 == 0 and n2 == 0
    assert np.array_equal(o[2], ca, equal_nan=True)
    _, half = _eval(L.half_eval, x)
    assert np.array_equal(o[3], half, equal_nan=True)          # device TC_POW_HALF == this host's pow(x,.5)


def _unary(f, which, x):
    x = np.ascontiguousarray(x)
    y = np.empty_like(x)
    dp = ctypes.POINTER(ctypes.c_double)
    rc = f(ctypes.c_int(which), x.ctypes.data_as(dp), y.ctypes.data_as(dp), ctypes.c_long(x.size))
    assert not (rc if f.__name__.startswith("crm_dev") else 0)          # (the host harness returns nothing)
    return y


def _libm_arguments(seed, n):
    """arguments for exp, log, sin, cos, atan: every branch of glibc's routines, the branch boundaries, special values"""
    r = np.random.RandomState(seed)
    sgn = lambda k: r.choice([-1.0, 1.0], k)           # noqa: E731
    xe = np.concatenate([r.uniform(-750, 710, n), r.normal(0, 1, n // 2), r.uniform(-1e-3, 1e-3, n // 8),
                         2.0 ** r.uniform(-1074, 10, n // 8) * sgn(n // 8),
                         np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 709.78, 709.79, 710.0, -708.4, -745.13, -745.14, -746.0, 1024.0,
                                   -1024.0, 1e300, -1e300, 5e-324, 1.0, -1.0, 2.0 ** -54, 2.0 ** -55, 511.9, 512.0, -512.0])])
    xl = np.concatenate([2.0 ** r.uniform(-1074, 1023, n), r.uniform(0.9, 1.1, n // 2), r.uniform(0, 4, n // 2),
                         1 + r.uniform(-2.0 ** -4, 2.0 ** -4, n // 4),
                         np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1.0, -1.0, 5e-324, 2.2250738585072014e-308,
                                   1.7976931348623157e308, 0.9375, 1.0644, 1.0645, 1.06445, 1.064697265625, -3e-320])])
    # sin / cos: Taylor (< 0.126), table (< 0.855469), pi/2 - x (< 2.426265), reduce_sincos (< 105414350), __branred beyond
    xt = np.concatenate([r.uniform(-0.126, 0.126, n // 4), r.uniform(-0.9, 0.9, n // 2), r.uniform(-2.5, 2.5, n // 2),
                         r.uniform(-100, 100, n // 2), r.uniform(-1.05e8, 1.05e8, n // 2), 2.0 ** r.uniform(-60, 26, n // 4) * sgn(n // 4),
                         r.uniform(1.05e8, 1e12, n // 4) * sgn(n // 4), 2.0 ** r.uniform(26.7, 1023.9, n // 2) * sgn(n // 2),
                         np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 0.126, 0.855469, 2.426265, 105414349.0, 105414350.0, 105414357.85,
                                   1e-300, 5e-324, np.pi, np.pi / 2, -np.pi / 2, 0.7853981633974483, 1e22, 1.7976931348623157e308,
                                   -1.7976931348623157e308, 2.0 ** 1000])])
    # atan: polynomial (< 1/16), table (< 1), through 1/x (< 16), pi/2 - 1/x (< 5.8e15), +-pi/2
    xa = np.concatenate([r.uniform(-1 / 16, 1 / 16, n // 4), r.uniform(-1, 1, n // 2), r.uniform(-16, 16, n // 2), r.uniform(-1000, 1000, n // 4),
                         2.0 ** r.uniform(-1074, 1023, n // 2) * sgn(n // 2),
                         np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1.0, -1.0, 1 / 16, 16.0, 5805361265115136.0, 5805361265115135.0,
                                   1e300, -1e300, 5e-324, 1.2904777690891933e-08, 0.99999999999, 15.999999])])
    return [("exp", xe), ("log", xl), ("sin", xt), ("cos", xt), ("atan", xa)]


def test_host_build_of_the_libm_restatements_is_the_hosts_libm():
    """csrc/tc_glibcpow.hip.inc: tc_glibc_exp / log / sin / cos / atan (what I_exp, I_log, I_sin, I_minus_sin, I_cos,
    I_minus_cos, I_atan and the srelu opcodes evaluate on the device, csrc/tc_pow.hip.inc) compiled for the host return the
    bits of THIS machine's libm: every binade, every branch of glibc's routines and their boundaries, overflow / underflow /
    subnormal results, arguments beyond 105 414 350 (the 1 800-bit reduction of branred.c), the special values"""
    L = build_host()
    for which, (name, x) in enumerate(_libm_arguments(3, 2_000_000)):
        a, b = _unary(L.unary_emul_eval, which, x), _unary(L.unary_libm_eval, which, x)
        assert np.array_equal(a, b, equal_nan=True), (name, int((~((a == b) | (np.isnan(a) & np.isnan(b)))).sum()))


@pytest.mark.gpu
def test_device_libm_functions_are_the_hosts_bit_for_bit():
    import torch  # noqa: F401  (one HIP runtime per process)
    L = build_host()
    D = ctypes.CDLL(build_dev())
    for which, (name, x) in enumerate(_libm_arguments(5, 4_000_000)):
        got, want = _unary(D.crm_dev_unary, which, x), _unary(L.unary_libm_eval, which, x)
        bad = int((~((got == want) | (np.isnan(got) & np.isnan(want)))).sum())
        print("\ndevice %s vs this host's on %d arguments: %d mismatches" % (name, x.size, bad))
        assert bad == 0, name
