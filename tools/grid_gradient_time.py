"""Cost of gpc_grid_gradient on one rank (1 x 1: replicate = copy, all tile columns local) against the single-GPU gradient.
usage: python tools/grid_gradient_time.py N D nb"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpc_amd import grid, synth  # noqa: E402
from gpc_amd.gp import CGp  # noqa: E402
import torch  # noqa: E402

N, D, nb = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
terms = [("rbf", [2.0 / D, 1.0]), ("white", [float(np.exp(-2.0))])]
X, y = synth.make_xy(N, D, 1234)
t1 = 0.0
if os.environ.get("GG_ONLY", "0") != "1":      # GG_ONLY=1: the grid's gradient alone (under a kernel trace)
    one = CGp(terms, X, y, ref_trans_rounding=False)
    one.logLikelihoodGradient()
    one._dirty()
    torch.cuda.synchronize()
    t0 = time.time()
    g1, ll1 = one.logLikelihoodGradient()
    torch.cuda.synchronize()
    t1 = time.time() - t0
    del one
    torch.cuda.empty_cache()
g = grid.create_local(1, 1, nb)[0]
g.set_problem(terms, X, y - y.mean(), None)
g.update_k()
g.gradient(3)
t0 = time.time()
g.update_k()
tf = time.time() - t0
t0 = time.time()
gg = g.gradient(3)
tg = time.time() - t0
print("N=%d D=%d nb=%d: single-GPU likelihood + gradient %.1f ms | grid 1x1: update_k %.1f ms + gradient %.1f ms (%.1f TFLOP/s of 2/3 N^3)"
      % (N, D, nb, t1 * 1e3, tf * 1e3, tg * 1e3, 2.0 * N ** 3 / 3.0 / tg * 1e-12))
