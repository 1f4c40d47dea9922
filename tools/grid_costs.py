"""Single-GPU kernel times the grid model (tools/grid_model.py) replays the scheduler's trace against: the staircase
trailing update as a function of its size, the panel's triangular solve, the diagonal tile's factorisation, device copies,
the cross-Gram build, the gap between two small launches.  Writes JSON (default profiles/r05_grid_costs.json).
usage (GPU box): python tools/grid_costs.py [out.json]"""
import ctypes
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpc_amd import _lib, api  # noqa: E402
import torch  # noqa: E402

out_path = sys.argv[1] if len(sys.argv) > 1 else "profiles/r05_grid_costs.json"
lib = _lib.load()
f = lib.gpc_bench_update
f.restype = ctypes.c_int
f.argtypes = [ctypes.c_int, ctypes.c_int64, ctypes.c_int64] + [ctypes.c_int] * 5 + [ctypes.POINTER(ctypes.c_double)] * 2


def timed(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(reps):
        fn()
    ev1.record()
    torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) / reps


res = {"device": api.device_info() if hasattr(api, "device_info") else {}, "update": [], "trsm_rlt": [], "potrf_tile": [],
       "copy": [], "gram_cross": []}

# ---- trailing update: a rank's share of an m x m lower triangle (2-D staircase), several grid shapes and sizes
for nb in (1024, 512):
    shapes = []
    for m in (2048, 4096, 8192, 16384, 32768, 65536):
        T = m // nb
        shapes += [(m, 2, 4, 0, 0), (m, 2, 4, 1, 3), (m, 8, 1, 0, 0), (m, 8, 1, 7, 0), (m, 4, 2, 0, 0), (m, 2, 2, 1, 0), (m, 2, 1, 0, 0)]
        if m <= 32768:
            shapes.append((m, 1, 1, 0, 0))
        # one tile column (U1 of a step): pc = T leaves every rank a single local tile column
        shapes += [(m, 2, T, 0, 0), (m, 8, T, 0, 0), (m, 1, T, 0, 0)]
    for (m, pr, pc, r, c) in shapes:
        ms, fl = ctypes.c_double(0), ctypes.c_double(0)
        rc = f(5, m, nb, pr, pc, r, c, 3, ctypes.byref(ms), ctypes.byref(fl))
        if rc != 0 or fl.value <= 0:
            continue
        T = m // nb
        rows = len(range(r, T, pr)) * nb
        cols = len(range(c, T, pc)) * nb
        res["update"].append({"nb": nb, "m": m, "pr": pr, "pc": pc, "r": r, "c": c, "rows": rows, "cols": cols,
                              "flops": fl.value, "ms": ms.value, "tflops": fl.value / ms.value * 1e-9})
        print("update nb=%d m=%d %dx%d (%d,%d) rows %d cols %d: %.3f ms %.1f TF" % (nb, m, pr, pc, r, c, rows, cols, ms.value,
                                                                                       fl.value / ms.value * 1e-9), flush=True)

# ---- panel solve B := B L^-T (the rows of a panel below its diagonal tile) and the tile factorisation
rng = np.random.RandomState(1)
for nb in (1024, 512):
    A = rng.randn(nb, nb) * 0.01
    A = A @ A.T + np.eye(nb)
    Ad = api.from_host(A)
    L = Ad.clone()
    api.potrf(L)

    def fac():
        L.copy_(Ad)
        api.potrf(L)

    def cp():
        L.copy_(Ad)

    t = timed(fac, reps=10) - timed(cp, reps=10)
    res["potrf_tile"].append({"nb": nb, "ms": t})
    print("potrf_tile nb=%d: %.3f ms" % (nb, t), flush=True)
    api.potrf(L)
    for M in (512, 1024, 2048, 4096, 8192, 16384, 32768, 65536):
        B = api.from_host(rng.randn(8, nb)).repeat(M // 8, 1).t().contiguous().t()      # M x nb, column-major
        B = api.empty(M, nb)
        B.normal_()
        t = timed(lambda: api.trsm(L, B, side="R", uplo="L", trans="T"), reps=5)
        res["trsm_rlt"].append({"nb": nb, "rows": M, "ms": t})
        print("trsm_rlt nb=%d rows=%d: %.3f ms" % (nb, M, t), flush=True)

# ---- a whole panel in one call (diagonal tile + the rows below it: one dataflow launch while it is short enough)
fp = lib.gpc_bench_panel
fp.restype = ctypes.c_int
fp.argtypes = [ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
res["potrf_panel"] = []
for nb in (1024, 512):
    for rows in (0, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536):
        ms = ctypes.c_double(0)
        _lib.check(fp(nb + rows, nb, 5, ctypes.byref(ms)))
        res["potrf_panel"].append({"nb": nb, "rows": nb + rows, "ms": ms.value})
        print("potrf_panel nb=%d rows=%d (+tile): %.3f ms" % (nb, rows, ms.value), flush=True)

# ---- device copies (the packing of row / column panels) and memsets
for mb in (1, 8, 32, 128, 512):
    n = mb * 1024 * 1024 // 8
    a = torch.empty(n, dtype=torch.float64, device="cuda").normal_()
    b = torch.empty_like(a)
    t = timed(lambda: b.copy_(a), reps=10)
    res["copy"].append({"bytes": 8 * n, "ms": t})
    print("copy %d MB: %.4f ms (%.0f GB/s moved)" % (mb, t, 16 * n / t * 1e-6), flush=True)

# ---- cross-Gram build of a rank's block
ks = api.kspec([("rbf", [2.0 / 32, 1.0])])
for (na, nbb, D) in ((8192, 8192, 32), (16384, 32768, 32), (8192, 65536, 32), (16384, 32768, 16)):
    Xa = api.from_host(rng.randn(na, D))
    Xb = api.from_host(rng.randn(nbb, D))
    K = api.empty(na, nbb)
    t = timed(lambda: api.gram_cross(ks, Xa, Xb, out=K), reps=3)
    res["gram_cross"].append({"rows": na, "cols": nbb, "D": D, "ms": t})
    print("gram_cross %d x %d D=%d: %.3f ms (%.0f GB/s written)" % (na, nbb, D, t, 8.0 * na * nbb / t * 1e-6), flush=True)
    del K

# ---- the gap between small dependent launches on one stream (what every scheduler step pays per kernel)
small = api.empty(128, 128)
small.zero_()
t = timed(lambda: api.add_diag_(small, 1.0), reps=200, warm=20)
res["small_launch_ms"] = t
print("small launch: %.4f ms" % t)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(200):
    api.add_diag_(small, 1.0)
res["host_issue_ms"] = (time.time() - t0) / 200 * 1e3
torch.cuda.synchronize()
print("host issue: %.4f ms" % res["host_issue_ms"])
os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
json.dump(res, open(out_path, "w"), indent=1)
print("wrote", out_path)
