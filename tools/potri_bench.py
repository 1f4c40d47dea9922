#!/usr/bin/env python
"""potrf vs potri vs the trsm pair on one size (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gpc_amd import api

def bench(fn, reps=2):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
D = 8
X = torch.randn((D, N), dtype=torch.float64, device="cuda").t()
ks = api.kspec([("rbf", [1.0, 1.0]), ("white", [0.1])])
K = api.empty(N, N); L = api.empty(N, N); W = api.empty(N, N)
api.gram_sym(ks, X, K)
def f_potrf():
    L.copy_(K); api.potrf(L, "L")
t_copy = bench(lambda: L.copy_(K))
f_potrf(); t_potrf = bench(f_potrf) - t_copy
def f_potri():
    W.copy_(L); api.potri(W, "L")
f_potri(); t_potri = bench(f_potri) - t_copy
y = torch.randn((1, N), dtype=torch.float64, device="cuda").t()
a = api.empty(N, 1)
def f_alpha():
    api.gp_alpha(L, y, out=a)
f_alpha(); t_alpha = bench(f_alpha, 5)
Y32 = torch.randn((32, N), dtype=torch.float64, device="cuda").t().clone() if False else torch.randn((32, N), dtype=torch.float64, device="cuda").t()
def f_trsm32():
    api.trsm(L, Y32, "L", "L", "N", "N")
f_trsm32(); t_trsm32 = bench(f_trsm32, 3)
print("N=%d potrf %.2f ms (%.1f TF)  potri %.2f ms (%.1f TF at 2N^3/3)  alpha(2 trsv) %.3f ms (%.0f GB/s of 2*4N^2)  trsm nrhs=32 %.3f ms"
      % (N, t_potrf, N**3 / 3 / t_potrf * 1e-9, t_potri, 2 * N**3 / 3 / t_potri * 1e-9, t_alpha, 8.0 * N * N / t_alpha * 1e-6, t_trsm32))
