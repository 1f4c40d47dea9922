import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpc_amd import api
v = int(sys.argv[1]) if len(sys.argv) > 1 else 2
M = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
K = int(sys.argv[3]) if len(sys.argv) > 3 else 512
api.check(api.lib().gpc_set_gemm_variant(v))
A = torch.randn((K, M), dtype=torch.float64, device="cuda").t()
C = torch.randn((M, M), dtype=torch.float64, device="cuda").t()
for _ in range(3):
    api.syrk(A, C, "L", "N", alpha=-1.0, beta=1.0)
torch.cuda.synchronize()
