"""How long does a small high-priority job (the factorisation of one nb x nb diagonal tile -- the head of the grid's panel
chain) take while a big trailing-update product runs on another stream?  Compared under GPC_GEMM_PF2=0 / 2.
usage: python tools/overlap_probe.py [M] [K] [nb]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from gpc_amd import api  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 24576
K = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
torch.manual_seed(0)
A = torch.randn((K, M), dtype=torch.float64, device="cuda").t()
C = torch.randn((M, M), dtype=torch.float64, device="cuda").t()
T0 = torch.randn((nb, nb), dtype=torch.float64, device="cuda")
T0 = (T0 @ T0.t() + nb * torch.eye(nb, dtype=torch.float64, device="cuda")).t().contiguous().t()
T = T0.clone()
big = torch.cuda.Stream(priority=0)
small = torch.cuda.Stream(priority=-1)


def run_big():
    with torch.cuda.stream(big):
        api.syrk(A, C, "L", "N", alpha=-1e-9, beta=1.0)


def run_small():
    with torch.cuda.stream(small):
        T.copy_(T0)
        api.potrf(T, "L")


def ms(fn, stream, reps=3):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(stream):
        e0.record()
    for _ in range(reps):
        fn()
    with torch.cuda.stream(stream):
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


t_big = ms(run_big, big)
t_small = ms(run_small, small, 5)
# both: the big product first, the small job 1 ms later on the high-priority stream
res = []
for _ in range(5):
    torch.cuda.synchronize()
    b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(big):
        b0.record()
    run_big()
    with torch.cuda.stream(big):
        b1.record()
    time.sleep(0.001)
    with torch.cuda.stream(small):
        s0.record()
    run_small()
    with torch.cuda.stream(small):
        s1.record()
    torch.cuda.synchronize()
    res.append((b0.elapsed_time(b1), s0.elapsed_time(s1), b0.elapsed_time(s1)))
res.sort(key=lambda r: r[1])
r = res[len(res) // 2]
print("PF2=%s M=%d K=%d nb=%d: product alone %.2f ms, tile factor alone %.3f ms | together: product %.2f ms, tile factor %.3f ms (done %.2f ms after the product started)"
      % (os.environ.get("GPC_GEMM_PF2", "default"), M, K, nb, t_big, t_small, r[0], r[1], r[2]))
