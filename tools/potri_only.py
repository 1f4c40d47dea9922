import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from gpc_amd import api
N = int(sys.argv[1])
X = torch.randn((8, N), dtype=torch.float64, device="cuda").t()
ks = api.kspec([("rbf", [1.0, 1.0]), ("white", [0.1])])
K = api.empty(N, N); api.gram_sym(ks, X, K); api.potrf(K, "L")
W = K.clone(); api.potri(W, "L"); torch.cuda.synchronize()
for _ in range(3):
    W.copy_(K); api.potri(W, "L")
torch.cuda.synchronize()
