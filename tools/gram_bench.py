import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gpc_amd import api, synth
N, D = int(sys.argv[1]) if len(sys.argv) > 1 else 32768, int(sys.argv[2]) if len(sys.argv) > 2 else 32
X = torch.randn((D, N), dtype=torch.float64, device="cuda").t()
ks = api.kspec([("rbf", [2.0 / D, 1.0]), ("white", [0.1])])
K = api.empty(N, N)
api.gram_sym(ks, X, K); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3): api.gram_sym(ks, X, K)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 3
print("N=%d D=%d debug=%s mfma=%s: %.3f ms  %.0f GB/s" % (N, D, os.environ.get("GPC_GRAM_DEBUG", "0"), os.environ.get("GPC_GRAM_MFMA", "1"), ms, 8.0 * N * N / ms * 1e-6))
