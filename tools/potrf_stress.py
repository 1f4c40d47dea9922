"""Random sizes through gpc_potrf_f64 (dataflow panels): factor against numpy, bit-identical on repetition, tall panels via
gpc_chol_inverse_f64.  usage: potrf_stress.py [count] [seed] [largest size, default 5000]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpc_amd import api
count = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
nmax = int(sys.argv[3]) if len(sys.argv) > 3 else 5000
worst = 0.0
for it in range(count):
    N = int(rng.choice([rng.randint(1, 200), rng.randint(200, 1500), rng.randint(1500, nmax)]))
    B = rng.randn(N, max(N // 3, 1))
    K = B @ B.T / max(N // 3, 1) + np.eye(N) * (0.1 + rng.rand(N))
    A1, A2 = api.from_host(K), api.from_host(K)
    assert api.potrf(A1, "L") == 0 and api.potrf(A2, "L") == 0
    L1, L2 = np.tril(api.to_host(A1)), np.tril(api.to_host(A2))
    assert np.array_equal(L1, L2), "not bit-identical at N=%d" % N
    Lr = np.linalg.cholesky(K)
    err = np.abs(L1 - Lr).max() / np.abs(Lr).max()
    worst = max(worst, err)
    assert err < 1e-11, (N, err)
    if N <= 3000:
        inv, logdet, info = api.chol_inverse(api.from_host(K))
        assert info == 0
        r = np.abs(api.to_host(inv) @ K - np.eye(N)).max()
        assert r < 1e-8, (N, r)
    print("N=%5d  rel err %.2e" % (N, err), flush=True)
print("ok: %d sizes, worst relative error %.2e" % (count, worst))
