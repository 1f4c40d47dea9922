"""The two triangular solves behind alpha = K^-1 m, timed apart (one right-hand side; forward L y = m, backward L' x = y).
usage (GPU box): python tools/trsv_bench.py [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpc_amd import api

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
X = torch.randn((8, N), dtype=torch.float64, device="cuda").t()
ks = api.kspec([("rbf", [1.0, 1.0]), ("white", [0.1])])
L = api.empty(N, N); api.gram_sym(ks, X, L); api.potrf(L, "L")
y = torch.randn((1, N), dtype=torch.float64, device="cuda").t()

def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

for name, tr in (("forward  L y = m ", "N"), ("backward L' x = y", "T")):
    t = timed(lambda: api.trsm(L, y, "L", "L", tr, "N"))
    print("N=%d %s %.3f ms  %.0f GB/s of 4N^2 bytes" % (N, name, t, 4.0 * N * N / t * 1e-6))

# a few dozen right-hand sides (alpha of a many-output GP): four at a time through the dataflow kernel (GPC_TRSM_GROUPS)
for nr in (8, 32, 64):
    Y = torch.randn((nr, N), dtype=torch.float64, device="cuda").t()
    t = timed(lambda: api.trsm(L, Y, "L", "L", "N", "N"), 2)
    print("N=%d forward solve, %d right-hand sides %.3f ms" % (N, nr, t))
