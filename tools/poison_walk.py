import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from gpc_amd import api, synth
N = int(sys.argv[1]); D = 32
X, y = synth.make_xy(N, D, 1234)
ks = api.kspec(synth.CONFIGS["cfg3"]["kern"])
Xd = api.from_host(X)
print("gram", flush=True); K = api.gram_sym(ks, Xd); torch.cuda.synchronize()
print("block", flush=True); blk = api.gram_block(ks, Xd, 3000, 300, 100, 200); torch.cuda.synchronize()
print("update_k", flush=True); L, logdet, jit, info = api.gp_update_k(ks, Xd, K); torch.cuda.synchronize(); print(info, logdet, flush=True)
print("alpha", flush=True); m = api.from_host(y - y.mean()); alpha = api.gp_alpha(L, m); torch.cuda.synchronize()
print("potri", flush=True); inv = L.clone(); api.potri(inv, "L"); torch.cuda.synchronize()
print("kern_grad", flush=True); g = api.kern_grad(ks, Xd, inv); print(g, flush=True)
print("done")
