#!/usr/bin/env python
"""Trailing-update time against m in steps of 128 at K = 1024 (N = 8192 view): where the rounds of workgroups show.
Run on the GPU box: python tools/syrk_small_m.py [mlo mhi K]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpc_amd import api
mlo = int(sys.argv[1]) if len(sys.argv) > 1 else 3584
mhi = int(sys.argv[2]) if len(sys.argv) > 2 else 7680
K = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
N = mhi + K
big = torch.zeros((N, N), dtype=torch.float64, device="cuda").t()
out = []
for m in range(mlo, mhi + 1, 128):
    k0 = N - m - K
    A = big[k0 + K:, k0:k0 + K]
    C = big[k0 + K:, k0 + K:]
    def fn(): api.syrk(A, C, "L", "N", alpha=-1e-12, beta=1.0)
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10
    tiles = (m // 128) * (m // 128 + 1) // 2
    out.append("m=%d tiles=%d (%.2f x 512): %.3f ms %.1f TF" % (m, tiles, tiles / 512.0, t, m * (m + 1) * K / t * 1e-9))
print("\n".join(out))
