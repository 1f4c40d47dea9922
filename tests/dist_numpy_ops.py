"""numpy stand-in for gpc_amd.dist.HipOps -- TEST INFRASTRUCTURE ONLY (like oracle/): it lets the CPU suite drive the
block-cyclic orchestration of gpc_amd/dist.py over gloo without a GPU.  The product never imports this file."""
import numpy as np
import scipy.linalg as sla
import torch


def kern(terms, X1, X2, sym_off=None):
    """k(X1_i, X2_j) for rbf / white / bias / lin terms; sym_off = (i0, j0) marks a block of the symmetric Gram
    (white lands where i0+i == j0+j), None a cross Gram (white contributes nothing)."""
    d2 = ((X1[:, None, :] - X2[None, :, :]) ** 2).sum(-1)
    K = np.zeros_like(d2)
    for name, p in terms:
        if name == "rbf":
            K += p[1] * np.exp(-0.5 * p[0] * d2)
        elif name == "bias":
            K += p[0]
        elif name == "lin":
            K += p[0] * (X1 @ X2.T)
        elif name == "white":
            if sym_off is not None:
                i = np.arange(X1.shape[0])[:, None] + sym_off[0]
                j = np.arange(X2.shape[0])[None, :] + sym_off[1]
                K += p[0] * (i == j)
        else:
            raise ValueError(name)
    return K


def kdiag(terms, X):
    d = np.zeros(X.shape[0])
    for name, p in terms:
        d += {"rbf": lambda: p[1], "bias": lambda: p[0], "white": lambda: p[0],
              "lin": lambda: p[0] * (X * X).sum(1)}[name]()
    return d


def _cm(rows, cols, fill=None):
    t = torch.empty((cols, rows), dtype=torch.float64) if fill is None else torch.full((cols, rows), fill,
                                                                                     dtype=torch.float64)
    return t.t()


class NumpyOps(object):
    def empty(self, rows, cols):
        return _cm(rows, cols, float("nan"))      # poison: reading uninitialised storage shows up in the results

    def zeros(self, rows, cols):
        return _cm(rows, cols, 0.0)

    def from_host(self, a):
        a = np.asarray(a, dtype=np.float64)
        return torch.from_numpy(np.ascontiguousarray(a.T)).t()

    def kspec(self, terms):
        return terms

    def info_word(self):
        return torch.zeros(1, dtype=torch.int32)

    def gram_block(self, ks, X, i0, m, j0, n, out):
        Xn = X.numpy()
        out.numpy()[...] = kern(ks, Xn[i0:i0 + m], Xn[j0:j0 + n], (i0, j0))

    def gram_cross(self, ks, X, X2, out):
        out.numpy()[...] = kern(ks, X.numpy(), X2.numpy(), None)

    def gram_diag(self, ks, X):
        return torch.from_numpy(kdiag(ks, X.numpy())).reshape(-1, 1)

    def potrf_panel(self, panel, col0, info):
        if int(info[0]) != 0:
            return
        a = panel.numpy()
        w = a.shape[1]
        A11 = np.tril(a[:w]) + np.tril(a[:w], -1).T          # only the lower triangle is defined
        try:
            L = np.linalg.cholesky(A11)
        except np.linalg.LinAlgError:
            info[0] = col0 + 1
            return
        low = np.tril_indices(w)
        a[:w][low] = L[low]
        if a.shape[0] > w:
            a[w:] = sla.solve_triangular(L, a[w:].T, lower=True).T

    def syrk_blockcyclic(self, P, C, row0, j0, pstride, nb):
        p, c = P.numpy(), C.numpy()
        M, ncols = c.shape
        assert p.shape[0] == M
        for c0 in range(0, ncols, nb):
            w = min(nb, ncols - c0)
            g0 = (j0 + (c0 // nb) * pstride) * nb
            upd = p @ p[g0 - row0:g0 - row0 + w].T
            rows = np.arange(M)[:, None] + row0
            cols = np.arange(w)[None, :] + g0
            c[:, c0:c0 + w] -= np.where(rows >= cols, upd, 0.0)

    def logdet_chol(self, Ljj):
        return float(2.0 * np.log(np.diag(Ljj.numpy())).sum())

    def colnorm2(self, A):
        return torch.from_numpy((A.numpy() ** 2).sum(0)).reshape(-1, 1)

    def gemm(self, A, B, C, transa, transb, alpha, beta):
        a = A.numpy().T if transa == "T" else A.numpy()
        b = B.numpy().T if transb == "T" else B.numpy()
        c = C.numpy()
        c[...] = alpha * (a @ b) + (beta * c if beta != 0.0 else 0.0)

    def trsm(self, A, B, trans):
        b = B.numpy()
        b[...] = sla.solve_triangular(np.tril(A.numpy()), b, lower=True, trans=1 if trans == "T" else 0)

    def trace(self, A):
        return float(np.trace(A.numpy()))

    def add_diag(self, A, c):
        a = A.numpy()
        a[np.diag_indices(min(a.shape))] += c
