#!/usr/bin/env python
"""SYRK rate vs trailing size m at K = 1024 (lower triangle of an m x m view inside an N x N matrix, ld = N): what the
trailing updates of one cfg-3 factorisation see.  Run on the GPU box."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpc_amd import api
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
K = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
big = torch.zeros((N, N), dtype=torch.float64, device="cuda").t()
tot_f, tot_t = 0.0, 0.0
for m in (63488, 57344, 49152, 40960, 32768, 24576, 16384, 8192, 4096):
    if m + K > N: continue
    k0 = N - m - K
    A = big[k0 + K:, k0:k0 + K]        # L21: m x K, ld N
    C = big[k0 + K:, k0 + K:]          # A22: m x m, ld N
    def fn(): api.syrk(A, C, "L", "N", alpha=-1.0, beta=1.0)
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); fn(); e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 2 * 1e-3
    print("m=%6d K=%d: %.3f ms %.2f TF/s" % (m, K, t * 1e3, m * (m + 1) * K / t * 1e-12))
