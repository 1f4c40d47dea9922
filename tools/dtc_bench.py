#!/usr/bin/env python
"""One DTC likelihood + gradient evaluation of the C++ CGp at scale (run on the GPU box): writes synthetic text inputs,
runs gp_hosttest dtc, reports its timed second evaluation."""
import os, subprocess, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpc_amd import synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
M = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
D = int(sys.argv[3]) if len(sys.argv) > 3 else 8
APPROX = [sys.argv[4]] if len(sys.argv) > 4 else []          # "dtcvar" | "fitc" (default DTC)
X, y = synth.make_xy(N, D, 3)
Xu = X[np.sort(np.random.RandomState(1).choice(N, M, replace=False))]
exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpc_amd", "host", "gp_hosttest")
with tempfile.TemporaryDirectory() as td:
    for nm, A in (("X", X), ("y", y), ("Xs", X[:4]), ("Xu", Xu)):
        np.savetxt(os.path.join(td, nm + ".txt"), A, fmt="%.17g")
    r = subprocess.run([exe, "dtc", td + "/X.txt", td + "/y.txt", td + "/Xs.txt", "rbf:0.25,1;white:0.01", td + "/Xu.txt", "100", "0"] + APPROX,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    out = r.stdout.decode()
    t = [ln for ln in out.splitlines() if ln.startswith("time_llgrad_ms") or ln.startswith("ll ")]
    print("N=%d M=%d D=%d %s" % (N, M, D, "".join(APPROX) or "dtc"), t, r.stderr.decode()[-300:])
