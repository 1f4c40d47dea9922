#!/usr/bin/env python
"""Rate of gpc_gemm_f64 in its four operand forms on square and skinny shapes (run on the GPU box)."""
import sys, os
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

for (M, N, K) in ((8192, 8192, 8192), (32768, 1024, 512), (1024, 65536, 1024), (16384, 16384, 512)):
    for ta, tb in (("N", "T"), ("N", "N"), ("T", "N"), ("T", "T")):
        A = torch.randn((K, M) if ta == "N" else (M, K), dtype=torch.float64, device="cuda").t()
        B = torch.randn((N, K) if tb == "N" else (K, N), dtype=torch.float64, device="cuda").t()
        C = torch.zeros((N, M), dtype=torch.float64, device="cuda").t()
        t = bench(lambda: api.gemm(A, B, C, ta, tb, alpha=1.0, beta=1.0))
        print("M=%6d N=%6d K=%5d %s%s : %8.3f ms  %6.2f TF/s" % (M, N, K, ta, tb, t * 1e3, 2.0 * M * N * K / t * 1e-12))
