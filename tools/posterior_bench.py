#!/usr/bin/env python
"""CGp::posteriorMeanVar at scale (run on the GPU box): N training points, Ns test points, factor given."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gpc_amd import api, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
Ns = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
D = 8
X, y = synth.make_xy(N, D, 3)
Xs = synth.make_xstar(Ns, D, 3)
ks = api.kspec([("rbf", [0.25, 1.0]), ("white", [0.01])])
Xd, yd, Xsd = api.from_host(X), api.from_host(y), api.from_host(Xs)
K = api.gram_sym(ks, Xd)
assert api.potrf(K, "L") == 0
a = api.empty(N, 1)
api.gp_alpha(K, yd, out=a)
def run():
    return api.gp_posterior(ks, Xd, K, a, Xsd)
mu, var = run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3): run()
e1.record(); torch.cuda.synchronize()
print("N=%d Ns=%d posterior mean+var %.3f ms   mu[0]=%.12g var[0]=%.12g" % (N, Ns, e0.elapsed_time(e1) / 3, float(mu[0, 0]), float(var[0])))
