import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gpc_amd import api
N, D = int(sys.argv[1]), int(sys.argv[2])
X = torch.randn((D, N), dtype=torch.float64, device="cuda").t()
for name, terms in (("rbf+white", [("rbf", [2.0 / D, 1.0]), ("white", [0.1])]),
                    ("rbfard+bias+white", [("rbfard", [2.0 / D, 1.0] + [0.5] * D), ("bias", [0.1]), ("white", [0.1])]),
                    ("rbf+rbf+lin", [("rbf", [2.0 / D, 1.0]), ("rbf", [0.5 / D, 0.3]), ("lin", [0.2]), ("white", [0.1])])):
    ks = api.kspec(terms)
    K = api.empty(N, N)
    api.gram_sym(ks, X, K); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): api.gram_sym(ks, X, K)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print("N=%d D=%d %-20s %.3f ms  %.0f GB/s" % (N, D, name, ms, 8.0 * N * N / ms * 1e-6))
