"""Few-right-hand-side triangular solves (the trsv path of gpc_trsm_f64) against scipy on ragged sizes."""
import numpy as np
import pytest
import scipy.linalg as sla


@pytest.mark.gpu
@pytest.mark.parametrize("M,d", [(1, 1), (63, 1), (64, 3), (65, 16), (200, 2), (1000, 12), (4096, 1), (4133, 5), (4133, 4), (8200, 3), (129, 2)])
@pytest.mark.parametrize("trans", ["N", "T"])
@pytest.mark.parametrize("diag", ["N", "U"])
def test_trsv_lower_vs_scipy(M, d, trans, diag):
    from gpc_amd import api
    rng = np.random.RandomState(M + d)
    A = rng.randn(M, M) * 0.3 / np.sqrt(M) + np.eye(M) * (1.0 + rng.rand(M))
    A = np.tril(A) + np.triu(rng.randn(M, M), 1)          # the upper part must be ignored
    B = rng.randn(M, d)
    Ad, Bd = api.from_host(A), api.from_host(B)
    api.trsm(Ad, Bd, side="L", uplo="L", trans=trans, diag=diag, alpha=0.7)
    want = sla.solve_triangular(np.tril(A), 0.7 * B, lower=True, trans=1 if trans == "T" else 0,
                                unit_diagonal=(diag == "U"))
    got = api.to_host(Bd)
    assert np.abs(got - want).max() <= 1e-11 * max(1.0, np.abs(want).max())
    # run-to-run bitwise reproducibility
    Bd2 = api.from_host(B)
    api.trsm(Ad, Bd2, side="L", uplo="L", trans=trans, diag=diag, alpha=0.7)
    assert np.array_equal(api.to_host(Bd2), got)
