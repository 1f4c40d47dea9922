cd $GRAFT_REPO_ROOT
for wgs in 256 512 1024; do for D in 32 8; do echo -n "lds_pad pair_wgs=$wgs "; GPC_GRAM_LDS_PAD=40000 GPC_GRAM_PAIR_WGS=$wgs python tools/gram_bench.py 65536 $D 2>/dev/null; done; done > gpurun_out/r13_gram.txt 2>&1
for D in 32 8; do echo -n "ldpad16 "; python - $D <<'PY' 2>/dev/null
import sys, os, torch
sys.path.insert(0, os.getcwd())
from gpc_amd import api
N, D = 65536, int(sys.argv[1])
X = torch.randn((D, N), dtype=torch.float64, device="cuda").t()
ks = api.kspec([("rbf", [2.0 / D, 1.0]), ("white", [0.1])])
K = api.empty(N + 16, N)[:N, :]
api.gram_sym(ks, X, K); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3): api.gram_sym(ks, X, K)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 3
print("N=%d D=%d ld=N+16: %.3f ms  %.0f GB/s" % (N, D, ms, 8.0 * N * N / ms * 1e-6))
PY
done >> gpurun_out/r13_gram.txt 2>&1
