"""A few launches of the rbfard parameter-gradient kernel on its own (for rocprofv3 --pmc): python tools/ard_grad_one.py N D [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from gpc_amd import api  # noqa: E402

N, D = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
X = torch.randn((D, N), dtype=torch.float64, device="cuda").t()
cg = torch.randn((N, N), dtype=torch.float64, device="cuda").t()
ks = api.kspec([("rbfard", [1.0, 1.0] + [0.5] * D), ("bias", [0.1]), ("white", [0.1])])
for _ in range(reps):
    api.kern_grad(ks, X, cg)
torch.cuda.synchronize()
