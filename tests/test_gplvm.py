"""GP-LVM objective (config 5): oracle restatement and HIP path against the compiled
reference's CGplvm on the oil data (goldens: tests/golden/gplvm_oil.npz, generator tests/golden/make_golden.py)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from make_golden import read_svml  # noqa: E402

E2 = float(np.exp(-2.0))
KERNS = {"ard": [("rbfard", [1.0, 1.0, 0.5, 0.5]), ("bias", [E2]), ("white", [E2])],
         "rbf": [("rbf", [1.0, 1.0]), ("bias", [E2]), ("white", [E2])],
         "lin": [("rbf", [2.0, 0.7]), ("lin", [0.3]), ("bias", [0.1]), ("white", [0.05])]}


@pytest.fixture(scope="module")
def oil(golden):
    Y, labs = read_svml(os.path.join(ROOT, "tests", "golden", "oilTrain.svml"))
    return Y, golden("gplvm_oil")


def _close(a, b, tol):
    a, b = np.asarray(a).ravel(), np.asarray(b).ravel()
    return np.abs(a - b).max() <= tol * max(1.0, np.abs(b).max())


@pytest.mark.parametrize("name", ["ard", "rbf", "lin"])
def test_oracle_gplvm_n200(oil, name):
    """the C restatement against the compiled reference: objective, log-det and all 406 gradient entries"""
    from oracle import portrun
    Y, g = oil
    for X, ll, gr in ((g["n200_X_pca"], g["n200_%s_ll" % name], g["n200_%s_g" % name]),
                      (g["n200_%s_Xp" % name], g["n200_%s_ll_p" % name], g["n200_%s_g_p" % name])):
        r = portrun.gplvm(KERNS[name], Y[:200], X)
        assert r["info"][0, 0] == 0
        assert abs(r["ll"][0, 0] - ll[0, 0]) <= 1e-10 * abs(ll[0, 0])
        assert _close(r["g"], gr, 1e-10)
    np.testing.assert_allclose(r["m"], g["n200_m"], rtol=0, atol=1e-14)


def test_pca_init_matches_reference_up_to_sign(oil):
    from gpc_amd.gplvm import pca_init
    Y, g = oil
    X = pca_init(Y[:200] - Y[:200].mean(axis=0, keepdims=True), 2)
    ref = g["n200_X_pca"]
    for c in range(2):
        s = np.sign(np.dot(X[:, c], ref[:, c]))
        np.testing.assert_allclose(s * X[:, c], ref[:, c], rtol=0, atol=1e-10)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["ard", "rbf", "lin"])
def test_hip_gplvm_n200(oil, name):
    from gpc_amd.gplvm import CGplvm
    Y, g = oil
    for X, ll, gr in ((g["n200_X_pca"], g["n200_%s_ll" % name], g["n200_%s_g" % name]),
                      (g["n200_%s_Xp" % name], g["n200_%s_ll_p" % name], g["n200_%s_g_p" % name])):
        mdl = CGplvm(KERNS[name], Y[:200], 2, X=X)
        grad, L = mdl.logLikelihoodGradient()
        assert abs(L - ll[0, 0]) <= 1e-8 * abs(ll[0, 0])
        assert _close(grad, gr, 1e-8)
        # parameter vector round trip
        p = mdl.getOptParams()
        mdl.setOptParams(p)
        assert abs(mdl.logLikelihood() - L) <= 1e-12 * abs(L)


@pytest.mark.gpu
def test_hip_gplvm_n1000_config5(oil):
    """full oil data: objective/gradient at the reference's PCA point and at its SCG end state (2006 parameters)"""
    from gpc_amd.gplvm import CGplvm
    Y, g = oil
    mdl = CGplvm(KERNS["ard"], Y, 2, X=g["n1000_X_pca"])
    grad, L = mdl.logLikelihoodGradient()
    assert abs(L - g["n1000_ll"][0, 0]) <= 1e-8 * abs(g["n1000_ll"][0, 0])
    assert _close(grad, g["n1000_g"], 1e-8)
    mdl.setOptParams(g["n1000_params_final"].ravel())
    assert abs(mdl.logLikelihood() - g["n1000_ll_final"][0, 0]) <= 1e-8 * abs(g["n1000_ll_final"][0, 0])
    np.testing.assert_allclose(mdl.X_host, g["n1000_X_final"], rtol=0, atol=1e-12)
    kf = [p for _, ps in mdl.terms for p in ps]
    np.testing.assert_allclose(kf, g["n1000_kern_final"].ravel(), rtol=1e-12)
