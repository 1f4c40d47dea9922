"""Cost of gpc_grid_gradient (distributed inverse + covGrad + kernel pass) on pr x pc thread ranks SHARING the one GPU
(correctness of the real kernels + exchange; the time is the sum of all ranks' work on one device).
usage: python tools/grid_gradient_shared.py N D nb pr pc"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpc_amd import grid, synth  # noqa: E402

N, D, nb, pr, pc = [int(v) for v in sys.argv[1:6]]
terms = [("rbf", [2.0 / D, 1.0]), ("white", [float(np.exp(-2.0))])]
X, y = synth.make_xy(N, D, 1234)
grids = grid.create_local(pr, pc, nb)


def work(g, rank):
    g.set_problem(terms, X, y - y.mean(), None)
    g.update_k()
    g.gradient(3)
    g.barrier()
    t0 = time.time()
    g.update_k()
    g.barrier()
    tf = time.time() - t0
    t0 = time.time()
    gg = g.gradient(3)
    g.barrier()
    return tf, time.time() - t0, gg, g.stats()


res = grid.run_local(grids, work)
tf, tg, gg, st = res[0]
print("N=%d D=%d nb=%d grid %dx%d (shared GPU): update_k %.1f ms, gradient %.1f ms (%.1f TFLOP/s of 2/3 N^3), held %.2f GB per rank, g=%s"
      % (N, D, nb, pr, pc, tf * 1e3, tg * 1e3, 2.0 * N ** 3 / 3.0 / tg * 1e-12, st["bytes_held"] * 1e-9, gg))
for g in grids:
    g.destroy()
