"""dpotri in place on a factor of N columns: time of the call alone (the copy that restores the factor is timed and subtracted).
usage (GPU box): python tools/potri_time.py [N]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpc_amd import api

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
X = torch.randn((8, N), dtype=torch.float64, device="cuda").t()
ks = api.kspec([("rbf", [1.0, 1.0]), ("white", [0.1])])
L = api.empty(N, N); api.gram_sym(ks, X, L); api.potrf(L, "L")
W = L.clone()

def timed(fn, reps):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

def call():
    W.copy_(L); api.potri(W, "L")
call()
t_copy = timed(lambda: W.copy_(L), 2)
t = timed(call, 2) - t_copy
print("N=%d potri %.1f ms = %.2f TFLOP/s at 2N^3/3" % (N, t, 2 * N**3 / 3 / t * 1e-9))
