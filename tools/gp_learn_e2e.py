#!/usr/bin/env python
"""End-to-end `gp learn` of the C++ host layer on a synthetic SVMlight file (run on the GPU box): wall time per SCG
iteration at a size where the O(N^3) pieces dominate."""
import os, subprocess, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpc_amd import synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
D = int(sys.argv[2]) if len(sys.argv) > 2 else 8
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
extra = sys.argv[4:]
X, y = synth.make_xy(N, D, 11)
exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpc_amd", "host", "gp")
with tempfile.TemporaryDirectory() as td:
    path = os.path.join(td, "data.svml")
    with open(path, "w") as f:
        for i in range(N):
            f.write("%.17g %s\n" % (y[i, 0], " ".join("%d:%.17g" % (q + 1, X[i, q]) for q in range(D))))
    t0 = time.time()
    r = subprocess.run([exe, "-v", "3", "-s", "1", "learn", "-#", str(iters)] + extra + [path, os.path.join(td, "m.model")],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    dt = time.time() - t0
    out = r.stdout.decode()
    keep = [ln for ln in out.splitlines() if ln.startswith(("Iteration", "Objective evaluations", "Log likelihood"))]
    print("N=%d D=%d iters=%d wall %.2f s rc=%d" % (N, D, iters, dt, r.returncode))
    print("\n".join(keep[-8:]))
    if r.returncode != 0:
        print(r.stderr.decode()[-500:])
