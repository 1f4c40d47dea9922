#!/usr/bin/env python
"""C++ host CGp (gp_hosttest gp) against the Python mirror and finite differences at scale (run on the GPU box)."""
import sys, os, subprocess, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpc_amd import gp, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
D = 8
X, y = synth.make_xy(N, D, 11)
e2 = float(np.exp(-2.0))
terms = [("rbf", [1.0, 1.0]), ("bias", [e2]), ("white", [e2])]
m = gp.CGp(terms, X, y)
g, ll = m.logLikelihoodGradient()
print("python ll %.12g grads %s" % (ll, g))
exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpc_amd", "host", "gp_hosttest")
with tempfile.TemporaryDirectory() as td:
    for nm, A in (("X", X), ("y", y), ("Xs", X[:4])):
        np.savetxt(os.path.join(td, nm + ".txt"), A, fmt="%.17g")
    r = subprocess.run([exe, "gp", td + "/X.txt", td + "/y.txt", td + "/Xs.txt", "rbf:1,1;bias:%.17g;white:%.17g" % (e2, e2)],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    for ln in r.stdout.decode().splitlines():
        if ln.split()[0] in ("ll", "ll_again", "grads", "ll_roundtrip", "logdet"):
            print("host  ", ln[:200])
    print(r.stderr.decode()[-300:])
