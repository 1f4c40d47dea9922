"""Overhead of the grid path on ONE rank (1 x 1 grid) against gpc_gp_update_k_f64 on the same inputs.
usage: python tools/grid_p1.py N D nb [reps]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpc_amd import api, grid, synth  # noqa: E402
import torch  # noqa: E402

N, D, nb = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
terms = [("rbf", [2.0 / D, 1.0]), ("white", [float(np.exp(-2.0))])]
X, y = synth.make_xy(N, D, 1234)
ks = api.kspec(terms)
Xd = api.from_host(X)
K = api.empty(N, N)
for _ in range(1):
    api.gp_update_k(ks, Xd, K)
torch.cuda.synchronize()
api.profile_enable(True)
t0 = time.time()
for _ in range(reps):
    _, ld1, _, info = api.gp_update_k(ks, Xd, K)
torch.cuda.synchronize()
t1 = (time.time() - t0) / reps
n1, ms1, fl1 = api.profile_read(0, reset=True)
del K
torch.cuda.empty_cache()
g = grid.create_local(1, 1, nb)[0]
g.set_lookahead(int(os.environ.get("GRID_LA", "1")))
g.set_problem(terms, X, y if os.environ.get("GRID_Y", "0") == "1" else None, None)
g.update_k()
t0 = time.time()
for _ in range(reps):
    ld2, _, info2 = g.update_k()
t2 = (time.time() - t0) / reps
n2, ms2, fl2 = api.profile_read(0, reset=True)
print("trailing updates: single %d launches %.2f ms %.1f TF | grid %d launches %.2f ms %.1f TF" % (
    n1 // reps, ms1 / reps, fl1 / ms1 * 1e-9, n2 // reps, ms2 / reps, fl2 / ms2 * 1e-9))
t0 = time.time()
for _ in range(reps):
    g.fill()
    g.sync()
tf = (time.time() - t0) / reps
fl = N ** 3 / 3.0
print("N=%d D=%d nb=%d  single-GPU update_k %.2f ms (%.1f TF)   grid 1x1 update_k %.2f ms (%.1f TF, fill %.2f ms)  overhead %.1f %%  "
      "logdet rel diff %.2e" % (N, D, nb, t1 * 1e3, fl / t1 * 1e-12, t2 * 1e3, fl / t2 * 1e-12, tf * 1e3, (t2 / t1 - 1) * 100,
                                abs(ld1 - ld2) / abs(ld1)))
