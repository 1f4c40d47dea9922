#!/usr/bin/env python
"""Finite-difference check of CGp::logLikelihoodGradient at scale (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpc_amd import gp, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
D = int(sys.argv[2]) if len(sys.argv) > 2 else 8
X, y = synth.make_xy(N, D, 11)
terms = [("rbf", [1.0, 1.0]), ("bias", [float(np.exp(-2.0))]), ("white", [float(np.exp(-2.0))])]   # gp learn's default kernel
m = gp.CGp(terms, X, y)
a0 = np.array(m.getOptParams(), dtype=float)
ll0 = m.logLikelihood()
g = np.array(m.logLikelihoodGradient()[0], dtype=float)
print("N=%d ll=%.10g" % (N, ll0))
print("analytic ", g)
fd = []
for i in range(len(a0)):
    h = 1e-5
    a = a0.copy(); a[i] += h; m.setOptParams(a); lp = m.logLikelihood()
    a = a0.copy(); a[i] -= h; m.setOptParams(a); lm = m.logLikelihood()
    fd.append((lp - lm) / (2 * h))
print("finite df", np.array(fd))
