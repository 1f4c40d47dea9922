"""Block-cyclic multi-GPU factorisation (gpc_amd/dist.py).

CPU: world_size 1/2/3 over gloo with a numpy stand-in for the local kernels (tests/dist_numpy_ops.py) -- checks the
orchestration: ownership, staircase indices, look-ahead order, extra-row forward substitution, back substitution.
GPU: the same job with the real HIP kernels, 1 rank and 2 ranks sharing the box's single GPU (gloo collectives,
host-staged), against the single-GPU library path and numpy.
"""
import os
import socket
import sys

import numpy as np
import pytest
import scipy.linalg as sla
import torch
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import dist_worker  # noqa: E402
from dist_numpy_ops import kern, kdiag  # noqa: E402

TERMS = [("rbf", [1.3, 0.9]), ("bias", [0.2]), ("white", [0.05])]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _expected(N, D, d, Ns):
    X, y, Xs = dist_worker.make_problem(N, D, d, Ns, 7)
    K = kern(TERMS, X, X, (0, 0))
    L = np.linalg.cholesky(K)
    z = sla.solve_triangular(L, y, lower=True)
    al = sla.solve_triangular(L, z, lower=True, trans=1)
    logdet = 2.0 * np.log(np.diag(L)).sum()
    ll = -0.5 * ((z * z).sum() + d * logdet) - d * N * 0.5 * np.log(2 * np.pi)
    out = {"L": L, "alpha": al, "logdet": logdet, "ll": ll}
    if Ns:
        ks = kern(TERMS, Xs, X, None)
        v = sla.solve_triangular(L, ks.T, lower=True)
        out["mu"] = ks @ al
        out["var"] = kdiag(TERMS, Xs) - (v * v).sum(0)
    return out


def _run(world, flavour, N, D, d, Ns, nb, tmp_path):
    port = _free_port()
    mp.spawn(dist_worker.run, args=(world, port, flavour, N, D, d, Ns, nb, TERMS, str(tmp_path)), nprocs=world,
             join=True)
    return [dict(np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))) for r in range(world)]


def _check(res, exp, Ns, tol):
    for r in res:
        assert abs(r["logdet"] - exp["logdet"]) <= tol * abs(exp["logdet"])
        assert abs(r["ll"] - exp["ll"]) <= tol * abs(exp["ll"])
        np.testing.assert_allclose(r["L"], exp["L"], rtol=0, atol=tol * np.abs(exp["L"]).max() * 10)
        np.testing.assert_allclose(r["alpha"], exp["alpha"], rtol=0, atol=tol * np.abs(exp["alpha"]).max() * 100)
        if Ns:
            np.testing.assert_allclose(r["mu"], exp["mu"], rtol=0, atol=tol * 100)
            np.testing.assert_allclose(r["var"], exp["var"], rtol=0, atol=tol * 100)
    # every rank reports the same replicated results
    for r in res[1:]:
        assert r["ll"] == res[0]["ll"]
        np.testing.assert_array_equal(r["alpha"], res[0]["alpha"])


@pytest.mark.parametrize("world,N,Ns,d,nb", [(1, 300, 5, 1, 128), (2, 700, 37, 2, 128), (3, 900, 0, 1, 128),
                                                (2, 512, 8, 1, 128), (2, 1100, 3, 2, 384), (3, 1300, 0, 1, 256)])
def test_blockcyclic_orchestration_gloo_cpu(world, N, Ns, d, nb, tmp_path):
    """ragged last panel, odd row count (padding row), more ranks than fit evenly, no test points; nb > 128: the
    panels are factored and broadcast in 128-column slabs (ragged last slab included)."""
    res = _run(world, "numpy", N, 3, d, Ns, nb, tmp_path)
    _check(res, _expected(N, 3, d, Ns), Ns, 1e-11)
    assert sum(int(r["ncols"]) for r in res) == N


def test_blockcyclic_jitter_schedule_gloo_cpu(tmp_path):
    """a singular Gram matrix (duplicated inputs, no white term): the distributed updateK walks jitChol's schedule and
    ends with the same total jitter on every rank as the single-matrix rule gives"""
    terms = [("rbf", [1.0, 1.0])]
    port = _free_port()
    mp.spawn(dist_worker.run_singular, args=(2, port, terms, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (dict(np.load(os.path.join(str(tmp_path), "sing%d.npz" % r))) for r in range(2))
    assert r0["jitter"] == r1["jitter"] and r0["jitter"] > 0
    X = dist_worker.singular_inputs()
    K = kern(terms, X, X, (0, 0))
    jit, total = 1e-6 * np.trace(K) / K.shape[0], 0.0
    while True:                                   # CMatrix::jitChol on the same matrix
        try:
            np.linalg.cholesky(K + total * np.eye(K.shape[0]))
            break
        except np.linalg.LinAlgError:
            total += jit
            jit *= 10.0
    assert abs(float(r0["jitter"]) - total) <= 1e-15 * max(total, 1.0)
    L = np.linalg.cholesky(K + total * np.eye(K.shape[0]))
    assert abs(float(r0["logdet"]) - 2 * np.log(np.diag(L)).sum()) <= 1e-6 * abs(2 * np.log(np.diag(L)).sum())


def test_index_helpers():
    from gpc_amd import dist as gdist
    g = gdist.DistGp.__new__(gdist.DistGp)
    g.nb, g.N, g.P = 128, 700, 3
    g.T = 6
    for rank in range(3):
        g.rank = rank
        mine = list(range(rank, g.T, g.P))
        for k in range(g.T):
            l0 = g.first_local_after(k)
            assert [j for j in mine if j > k] == mine[l0:]
    assert g.width(5) == 60 and g.width(0) == 128 and g.owner(4) == 1 and g.lcol(4) == 128


def test_hipops_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from gpc_amd import dist as gdist
    with pytest.raises(RuntimeError):
        gdist.HipOps()


@pytest.mark.gpu
@pytest.mark.parametrize("world,N,Ns,d,nb", [(1, 1500, 33, 2, 128), (2, 1500, 33, 2, 128), (2, 4096, 0, 1, 512),
                                             (2, 2049, 7, 1, 256)])
def test_blockcyclic_hip(world, N, Ns, d, nb, tmp_path):
    res = _run(world, "hip" if nb != 256 else "hip-sync", N, 3, d, Ns, nb, tmp_path)
    _check(res, _expected(N, 3, d, Ns), Ns, 1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("flavour", ["hip-rccl", "hip-rccl-sync"])
def test_blockcyclic_rccl_calls_single_rank(flavour, tmp_path):
    """The RCCL code path (async broadcast on the panel stream, work.wait(), int64 MIN / fp64 SUM all-reduce,
    high-priority communicator) in a 1-rank job: all a 1-GPU box can run of it -- and the serialised fall-back mode
    (blocking broadcasts on the main stream) bench.py switches to if its start-up self-check fails."""
    res = _run(1, flavour, 1500, 3, 2, 33, 128, tmp_path)
    _check(res, _expected(1500, 3, 2, 33), 33, 1e-9)


@pytest.mark.gpu
def test_staircase_update_vs_numpy():
    """gpc_syrk_blockcyclic_f64 alone: every (row0, j0, pstride) combination a 3-rank job meets."""
    from gpc_amd import api
    from dist_numpy_ops import NumpyOps
    rng = np.random.RandomState(3)
    nb, Pn = 128, 3
    for (Mtot, k, rank, npan, lastw) in [(1100, 0, 1, 3, 128), (1100, 1, 0, 2, 60), (2000, 2, 2, 4, 128)]:
        row0 = (k + 1) * nb
        M = Mtot - row0
        l0 = 0 if k < rank else (k - rank) // Pn + 1
        j0 = rank + l0 * Pn
        ncols = (npan - 1) * nb + lastw
        Ph = rng.randn(M, nb)
        Ch = rng.randn(M, ncols)
        Pd, Cd = api.from_host(Ph), api.from_host(Ch)
        api.syrk_blockcyclic(Pd, Cd, row0, j0, Pn, nb)
        Pt = torch.from_numpy(np.ascontiguousarray(Ph.T)).t()
        Ct = torch.from_numpy(np.ascontiguousarray(Ch.T)).t()
        NumpyOps().syrk_blockcyclic(Pt, Ct, row0, j0, Pn, nb)
        got, want = api.to_host(Cd), Ct.numpy()
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-11 * nb)
        # entries above the global diagonal are untouched, bit for bit
        rows = np.arange(M)[:, None] + row0
        gc = np.array([(j0 + (c // nb) * Pn) * nb + c % nb for c in range(ncols)])[None, :]
        assert np.array_equal(got[rows < gc], Ch[rows < gc])
