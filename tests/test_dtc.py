"""Sparse approximation DTC: oracle restatement and HIP path against the compiled reference's
CGp(approxType = DTC) (goldens: tests/golden/gp_dtc.npz, generator tests/golden/make_golden.py dtc)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gpc_amd import synth  # noqa: E402

E2 = float(np.exp(-2.0))
CASES = {"a": [("rbf", [1.0, 1.0]), ("bias", [E2]), ("white", [E2])],
         "b": [("rbfard", [1.3, 0.8, 0.3, 0.9, 0.5]), ("lin", [0.2]), ("white", [0.05])],
         "c": [("rbf", [0.25, 1.0]), ("white", [0.01])]}


def problem(g, name):
    N, D = int(g[name + "_N"]), int(g[name + "_D"])
    X, y = synth.make_xy(N, D, seed=5)
    return X, y, g[name + "_Xu"], float(g[name + "_beta"]), g[name + "_Xstar"]


def close(a, b, tol):
    a, b = np.asarray(a).ravel(), np.asarray(b).ravel()
    return np.abs(a - b).max() <= tol * max(1.0, np.abs(b).max())


@pytest.mark.parametrize("name", ["a", "b", "c"])
def test_oracle_dtc(golden, name):
    from oracle import refrun
    g = golden("gp_dtc")
    X, y, Xu, beta, Xs = problem(g, name)
    arr = dict(refrun.kern_arrays(CASES[name]))
    arr.update({"X": X, "y": y, "X_u": Xu, "beta": beta, "Xstar": Xs})
    r = refrun.run_port("dtc", arr)
    assert r["info"][0, 0] == 0
    assert abs(r["ll"][0, 0] - g[name + "_ll"][0, 0]) <= 1e-8 * abs(g[name + "_ll"][0, 0])
    assert close(r["grads"], g[name + "_grads"], 1e-8)
    assert close(r["alpha"], g[name + "_alpha"], 1e-8)
    assert close(r["mu"], g[name + "_mu"], 1e-8)
    assert close(r["var"], g[name + "_var"][:, :1], 1e-8)
