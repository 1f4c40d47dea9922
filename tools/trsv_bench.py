#!/usr/bin/env python
"""The two triangular solves of CGp::updateAlpha on a synthetic lower factor (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpc_amd import api
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
d = int(sys.argv[2]) if len(sys.argv) > 2 else 1
L = torch.rand((N, N), dtype=torch.float64, device="cuda").t() * (0.5 / N)
L.diagonal().fill_(1.0)
y = torch.randn((d, N), dtype=torch.float64, device="cuda").t()
a = api.empty(N, d)
api.gp_alpha(L, y, out=a); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3): api.gp_alpha(L, y, out=a)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 3
print("N=%d d=%d alpha (2 solves) %.3f ms  %.0f GB/s algorithmic (2 x 4N^2 bytes)" % (N, d, ms, 8.0 * N * N / ms * 1e-6))
