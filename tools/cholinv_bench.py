"""gpc_chol_inverse_f64 (one call: factor + inverse + log-det) against gpc_potrf_f64 + gpc_potri_f64, per N.  GPC_CHOLINV_MAXN moves
the size up to which the augmented [K; I] factorisation is used."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gpc_amd import api, synth
for N in [int(a) for a in sys.argv[1:]] or [1000, 2048, 4096, 8192]:
    X, _ = synth.make_xy(N, 8, 1234)
    ks = api.kspec([("rbf", [1.0, 1.0]), ("white", [0.01])])
    Xd = api.from_host(X)
    K0 = api.empty(N, N)
    api.gram_sym(ks, Xd, K0)
    def run():
        K = K0.clone()
        return api.chol_inverse(K)
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): run()
    e1.record(); torch.cuda.synchronize()
    print("N=%d chol_inverse (GPC_CHOLINV_MAXN=%s): %.3f ms" % (N, os.environ.get("GPC_CHOLINV_MAXN", "2048"), e0.elapsed_time(e1) / 5))
