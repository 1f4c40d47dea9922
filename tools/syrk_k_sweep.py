#!/usr/bin/env python
"""SYRK rate vs K and beta on one big shape: separates the per-tile fixed cost (prologue + C read-modify-write) from the
steady-state loop rate of gemm_nt_fast_kernel.  Run on the GPU box."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gpc_amd import api

def bench(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3

M = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
torch.manual_seed(0)
C = torch.randn((M, M), dtype=torch.float64, device="cuda").t()
for K in (128, 256, 512, 1024, 2048, 4096):
    A = torch.randn((K, M), dtype=torch.float64, device="cuda").t()
    for beta in (1.0, 0.0):
        t = bench(lambda: api.syrk(A, C, "L", "N", alpha=-1.0, beta=beta))
        print("M=%d K=%4d beta=%.0f : %.3f ms  %.2f TF/s" % (M, K, beta, t * 1e3, M * (M + 1) * K / t * 1e-12))
