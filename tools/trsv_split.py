import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from gpc_amd import api
for N in (8192, 65536):
    L = torch.rand((N, N), dtype=torch.float64, device="cuda").t() * (0.5 / N)
    L.diagonal().fill_(1.0)
    y = torch.randn((1, N), dtype=torch.float64, device="cuda").t()
    for tr in ("N", "T"):
        b = y.clone()
        api.trsm(L, b, "L", "L", tr, "N", 1.0); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): api.trsm(L, b, "L", "L", tr, "N", 1.0)
        e1.record(); torch.cuda.synchronize()
        print("N=%d trans=%s: %.3f ms" % (N, tr, e0.elapsed_time(e1) / 3))
