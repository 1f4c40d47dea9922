import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gpc_amd import api
api.lib()
def t(fn, reps=50):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for lda in (64, 1024, 8192, 65536):
    big = torch.zeros((64, lda), dtype=torch.float64, device="cuda")   # 64 columns of length lda
    A = big.t()[:64, :64]      # 64 x 64 view with leading dimension lda
    base = (torch.eye(64, dtype=torch.float64, device="cuda") * 64 + 1.0)
    def run():
        A.copy_(base)
        api.potrf(A, "L")
    def run_copy():
        A.copy_(base); 
    print("lda %6d : potrf(64) incl. info readback %.1f us, copy only %.1f us" % (lda, t(run), t(run_copy)))
