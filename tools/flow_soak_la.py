"""Soak of the look-ahead factorisation (dataflow panels on the panel stream beside the trailing updates): the factor of the same
matrix, bit for bit, many times.  usage: flow_soak_la.py N reps"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gpc_amd import api, synth
N, reps = int(sys.argv[1]), int(sys.argv[2])
X, _ = synth.make_xy(N, 8, 99)
ks = api.kspec([("rbf", [1.0, 1.0]), ("white", [0.05])])
Xd = api.from_host(X)
K = api.empty(N, N)
L, ld0, jit, info = api.gp_update_k(ks, Xd, K)
ref = torch.tril(L).clone()
bad = 0
for r in range(reps):
    L, ld, jit, info = api.gp_update_k(ks, Xd, K)
    if info != 0 or ld != ld0 or not torch.equal(torch.tril(L), ref):
        bad += 1
        print("MISMATCH at rep", r, info, ld, ld0, flush=True)
print("N=%d: %d repetitions, mismatches: %d, logdet %.17g" % (N, reps, bad, ld0))
sys.exit(1 if bad else 0)
