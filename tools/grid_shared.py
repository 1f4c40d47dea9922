"""pr x pc ranks as threads SHARING one GPU: the whole job's time against the single-GPU factorisation of the same matrix.
The arithmetic is conserved, so what this prices is the driver itself -- tile bookkeeping, packing, the exchange as
device-to-device copies, the loss from smaller launches -- not xGMI.  usage: python tools/grid_shared.py N D nb [pr pc ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpc_amd import api, grid, synth  # noqa: E402
import torch  # noqa: E402

N, D, nb = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
shapes = [(int(sys.argv[i]), int(sys.argv[i + 1])) for i in range(4, len(sys.argv) - 1, 2)] or [(1, 1), (1, 2), (2, 2), (2, 4)]
terms = [("rbf", [2.0 / D, 1.0]), ("white", [float(np.exp(-2.0))])]
X, y = synth.make_xy(N, D, 1234)
ks = api.kspec(terms)
Xd = api.from_host(X)
K = api.empty(N, N)
api.gp_update_k(ks, Xd, K)
torch.cuda.synchronize()
t0 = time.time()
_, ld1, _, _ = api.gp_update_k(ks, Xd, K)
torch.cuda.synchronize()
t1 = time.time() - t0
del K
torch.cuda.empty_cache()
print("N=%d D=%d nb=%d   single GPU %.1f ms" % (N, D, nb, t1 * 1e3))
for pr, pc in shapes:
    gs = grid.create_local(pr, pc, nb)

    def work(g, rank):
        g.set_problem(terms, X, None, None)
        g.update_k()
        g.barrier()
        t0 = time.time()
        ld, _, info = g.update_k()
        dt = time.time() - t0
        st = g.stats()
        return dt, ld, info, st

    res = grid.run_local(gs, work)
    for g in gs:
        g.destroy()
    dt = max(r[0] for r in res)
    print("  %d x %d on one device: %.1f ms (x%.2f of single)  logdet rel diff %.1e  rank0 received %.2f GB (row) + %.2f GB (col)"
          % (pr, pc, dt * 1e3, dt / t1, abs(res[0][1] - ld1) / abs(ld1), res[0][3]["bytes_row"] / 2 * 1e-9,
             res[0][3]["bytes_col"] / 2 * 1e-9))
