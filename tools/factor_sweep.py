"""Time of one gpc_gp_update_k_f64 (Gram + Cholesky + log-det) over a range of sizes, for A/B runs of environment switches:
usage: [ENV=...] python tools/factor_sweep.py N [N ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from gpc_amd import api, synth  # noqa: E402

out = []
for N in [int(a) for a in sys.argv[1:]]:
    X, _ = synth.make_xy(N, 8, 1234)
    ks = api.kspec([("rbf", [2.0 / 8, 1.0]), ("white", [0.05])])
    Xd = api.from_host(X)
    K = api.empty(N, N)
    reps = 3 if N >= 32768 else 10
    for _ in range(2):
        api.gp_update_k(ks, Xd, K)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        _, ld, _, info = api.gp_update_k(ks, Xd, K)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    out.append("N=%d %.2f ms (%.1f TF)" % (N, dt * 1e3, N ** 3 / 3.0 / dt * 1e-12))
    del K
    torch.cuda.empty_cache()
print(" | ".join(out))
