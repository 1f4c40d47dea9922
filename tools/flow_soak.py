"""Soak run for the dataflow kernels: the same matrices factored / solved many times, every result compared bit for bit with the
first one (a missed or torn value in an exchange buffer would show up as a difference).  usage: flow_soak.py [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gpc_amd import api
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.RandomState(3)
cases = []
for N in (257, 1000, 2048, 3333, 4096, 8192):
    B = rng.randn(N, max(N // 3, 1))
    K = B @ B.T / max(N // 3, 1) + np.eye(N) * (0.1 + rng.rand(N))
    Kd = api.from_host(K)
    y = api.from_host(rng.randn(N, 2))
    A = Kd.clone()
    assert api.potrf(A, "L") == 0
    al = api.gp_alpha(A, y)
    cases.append((N, Kd, y, A.clone(), al.clone()))
bad = 0
for r in range(reps):
    for N, Kd, y, Aref, alref in cases:
        A = Kd.clone()
        assert api.potrf(A, "L") == 0
        al = api.gp_alpha(A, y)
        if not (torch.equal(torch.tril(A), torch.tril(Aref)) and torch.equal(al, alref)):
            bad += 1
            print("MISMATCH at rep %d N=%d" % (r, N), flush=True)
    if r % 50 == 49: print("rep", r + 1, "ok so far" if bad == 0 else "mismatches: %d" % bad, flush=True)
print("soak:", reps, "repetitions x", len(cases), "sizes, mismatches:", bad)
sys.exit(1 if bad else 0)
