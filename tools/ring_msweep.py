"""Trailing-update rate (lower SYRK-shaped NT product, beta = 1) over M at the factorisation's panel widths, under whatever
GPC_GEMM_RING says.  usage: python tools/ring_msweep.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpc_amd import api

def bench(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3

out = []
for M in (4096, 6144, 8192, 10240, 12288, 16384, 20480, 24576, 32768, 49152):
    C = torch.randn((M, M), dtype=torch.float64, device="cuda").t()
    for K in (1024, 1536):
        A = torch.randn((K, M), dtype=torch.float64, device="cuda").t()
        t = bench(lambda: api.syrk(A, C, "L", "N", alpha=-1.0, beta=1.0))
        out.append("M=%5d K=%4d %.3f ms %.2f TF/s" % (M, K, t * 1e3, M * (M + 1) * K / t * 1e-12))
    del C
print("\n".join(out))
