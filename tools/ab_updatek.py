"""A/B of gpc_gp_update_k_f64 between two builds of the library ON THE SAME BOX, interleaved (box-to-box clocks differ by
several per cent, so only same-box comparisons mean anything).  usage: python tools/ab_updatek.py libA.so libB.so N D [reps]"""
import ctypes
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from gpc_amd import _lib, synth  # noqa: E402

pa, pb, N, D = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
libs = [ctypes.CDLL(pa), ctypes.CDLL(pb)]
sig = _lib.SIGNATURES["gpc_gp_update_k_f64"]
for L in libs:
    L.gpc_gp_update_k_f64.restype, L.gpc_gp_update_k_f64.argtypes = sig
ks = _lib.KSpec()
ks.n_terms = 2
ks.types[0], ks.types[1] = 1, 3
ks.offs[0], ks.offs[1], ks.offs[2] = 0, 2, 3
ks.params[0], ks.params[1], ks.params[2] = 2.0 / D, 1.0, float(np.exp(-2.0))
X, _ = synth.make_xy(N, D, 1234)
Xd = torch.from_numpy(np.ascontiguousarray(X.T)).cuda()
K = torch.empty((N, N), dtype=torch.float64, device="cuda")
ld, jit, info = ctypes.c_double(), ctypes.c_double(), ctypes.c_int()


def run(L):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rc = L.gpc_gp_update_k_f64(ctypes.byref(ks), Xd.data_ptr(), N, D, N, K.data_ptr(), N, ctypes.byref(ld), ctypes.byref(jit),
                               ctypes.byref(info), None)
    torch.cuda.synchronize()
    assert rc == 0 and info.value == 0
    return time.perf_counter() - t0


for L in libs:
    run(L)
ts = [[], []]
for _ in range(reps):
    for i, L in enumerate(libs):
        ts[i].append(run(L))
for i, p in enumerate((pa, pb)):
    t = min(ts[i])
    print("%-40s best %.2f ms  median %.2f ms  (%.1f TF)" % (p[-40:], t * 1e3, float(np.median(ts[i])) * 1e3, N ** 3 / 3.0 / t * 1e-12))
