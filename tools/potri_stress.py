"""Random sizes through gpc_potri_f64 and the side-R solve (dataflow launches in "given" mode): against numpy, bit-identical
on repetition.  usage: potri_stress.py [count] [seed] [largest size]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from gpc_amd import api  # noqa: E402

count = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
nmax = int(sys.argv[3]) if len(sys.argv) > 3 else 5000
worst = 0.0
for it in range(count):
    N = int(rng.choice([rng.randint(2, 300), rng.randint(300, 2000), rng.randint(2000, nmax)]))
    B = rng.randn(N, max(N // 3, 1))
    K = B @ B.T / max(N // 3, 1) + np.eye(N) * (0.5 + rng.rand(N))
    Lh = np.linalg.cholesky(K)
    a1, a2 = api.from_host(Lh), api.from_host(Lh)
    api.potri(a1, "L")
    api.potri(a2, "L")
    i1, i2 = api.to_host(a1), api.to_host(a2)
    assert np.array_equal(i1, i2), "inverse not bit-identical at N=%d" % N
    r = np.abs(i1 @ K - np.eye(N)).max()
    M = int(rng.randint(1, 700))
    R = rng.randn(M, N)
    b1 = api.from_host(R)
    api.trsm(api.from_host(Lh), b1, "R", "L", "T", "N")
    X = api.to_host(b1)
    r2 = np.abs(X @ Lh.T - R).max() / np.abs(R).max()
    worst = max(worst, r, r2)
    assert r < 1e-9 and r2 < 1e-10, (N, M, r, r2)
    print("N=%5d M=%4d  |K^-1 K - I| %.2e  |X L' - B| %.2e" % (N, M, r, r2), flush=True)
print("ok: %d sizes, worst %.2e" % (count, worst))
