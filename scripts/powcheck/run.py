import ctypes, numpy as np, os
here = os.path.dirname(os.path.abspath(__file__))
import torch  # noqa: F401  (one HIP runtime in the process)
D = ctypes.CDLL(os.path.join(here, "libpowcheck.so")); H = ctypes.CDLL(os.path.join(here, "libhostpow.so"))
n = 1 << 20
x = np.random.RandomState(0).uniform(1e-3, 500, n)
dp = ctypes.POINTER(ctypes.c_double)
P = lambda a: a.ctypes.data_as(dp)
a, b, c, d = (np.empty(n) for _ in range(4)); ha, hb, hc = (np.empty(n) for _ in range(3))
print("rc", D.run(P(x), P(a), P(b), P(c), P(d), n)); H.hostpow(P(x), P(ha), P(hb), P(hc), n)
for nm, u, v in (("Dsqrt", a, ha), ("DDsqrt", b, hb), ("sqrt", c, hc), ("0.5/sqrt vs host Dsqrt", d, ha)):
    ulp = np.abs(u - v) / np.spacing(np.abs(v))
    print(f"{nm}: mismatching {np.mean(u != v)*100:.3f}%  max ulp {ulp.max():.1f}")
