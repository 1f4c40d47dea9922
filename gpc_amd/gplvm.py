"""Python mirror of the reference's CGplvm (plain model: no dynamics, no back constraints, output scales not learnt) on
top of the C-ABI.  Method names follow CGplvm.cpp so that the parity tests read like the reference.

Per objective evaluation (CGplvm::logLikelihoodGradient, CGplvm.cpp:555-716) the device does, all in libgpc_hip.so:
  Gram of the latent points (gpc_gram_sym_f64) -> Cholesky (gpc_potrf_f64; no jitter, as CGplvm::_updateInvK calls
  plain chol(), CGplvm.cpp:435-446) -> log-det -> inverse (gpc_potri_f64) -> A = invK m (gpc_gemm_f64) ->
  G = sum_j covGrad_j (gpc_covgrad_multi_f64) -> kernel-parameter gradient (gpc_kern_grad_f64) and dL/dX
  (gpc_kern_gradx_f64), each ONE pass over G instead of the reference's d passes.
The PCA initialisation (CGplvm::initXpca, CGplvm.cpp:157-192) is an eigen-decomposition of the d x d data covariance
and stays on the host, as SURVEY.md section 8f says.
"""
import numpy as np

from . import api
from .gp import _atox, _gradfact, _xtoa, param_transforms


def pca_init(m, q):
    """CGplvm::initXpca (non back-constrained branch): X = m U_q diag(lambda_q)^-1/2, columns centred.
    The sign of each eigenvector is LAPACK's business in the reference; here the largest-magnitude component of every
    eigenvector is made positive (the objective is invariant to the sign of a latent column)."""
    N = m.shape[0]
    ymean = m.mean(axis=0, keepdims=True)
    cov = m.T @ m / float(N) - ymean.T @ ymean
    w, U = np.linalg.eigh(cov)
    X = np.empty((N, q))
    for i in range(q):
        u = U[:, -1 - i].copy()
        if u[np.argmax(np.abs(u))] < 0:
            u = -u
        X[:, i] = m @ (u / np.sqrt(w[-1 - i]))
    return X - X.mean(axis=0, keepdims=True)


class CGplvm:
    def __init__(self, terms, Y, latent_dim=2, regularise=True, X=None, device="cuda"):
        self.terms = [(n, list(map(float, p))) for n, p in terms]
        self.kinds = param_transforms(self.terms)
        Y = np.asarray(Y, dtype=np.float64)
        self.N, self.d = Y.shape
        self.q = int(latent_dim)
        self.device = device
        self.regularise = bool(regularise)
        self.bias = Y.mean(axis=0)                       # CScaleNoise: bias = meanCol(y), scale 1 (gplvm.cpp:498-507)
        self.m_host = Y - self.bias[None, :]             # CScaleNoise::updateSites, CNoise.cpp:710-721
        self.m = api.from_host(self.m_host, device)
        self.X_host = pca_init(self.m_host, self.q) if X is None else np.array(X, dtype=np.float64)
        self.X = api.from_host(self.X_host, device)
        self.logDetK = None

    # ---- parameters: [kernel (transformed)..., X column by column], CGplvm.cpp:257-330 -------------------------------
    def _flat(self):
        return [p for _, ps in self.terms for p in ps]

    def getOptNumParams(self):
        return len(self.kinds) + self.N * self.q

    def getOptParams(self):
        k = [_xtoa(kind, x) for kind, x in zip(self.kinds, self._flat())]
        return np.concatenate([np.array(k), self.X_host.reshape(-1, order="F")])

    def setOptParams(self, a):
        nk = len(self.kinds)
        it = iter([_atox(kind, float(v)) for kind, v in zip(self.kinds, a[:nk])])
        self.terms = [(n, [next(it) for _ in ps]) for n, ps in self.terms]
        self.X_host = np.asarray(a[nk:], dtype=np.float64).reshape(self.N, self.q, order="F").copy()
        self.X = api.from_host(self.X_host, self.device)

    # ---- objective and gradient ----------------------------------------------------------------------------------------
    def logLikelihoodGradient(self, want_grad=True):
        """Returns (g, ll): CGplvm::logLikelihoodGradient (g = None when want_grad is False: CGplvm::logLikelihood)."""
        ks = api.kspec(self.terms)
        N, d, q = self.N, self.d, self.q
        K = api.gram_sym(ks, self.X)                                  # _updateK, CGplvm.cpp:418-432
        # _updateInvK: chol() (no jitter here), logDet, pdinv in one pass (CGplvm.cpp:441-444)
        invK, self.logDetK, info = api.chol_inverse(K)
        if info != 0:
            raise np.linalg.LinAlgError("MatrixNonPosDef: leading minor %d" % info)
        A = api.zeros(N, d, self.device)
        api.gemm(invK, self.m, A)                                     # invK * m  (dsymv per column in the reference)
        quad = api.coldot(A, self.m)
        L = 0.0
        for j in range(d):                                            # CGplvm.cpp:498-507
            L += quad[j]
            L += self.logDetK
        if self.regularise:
            L += float(api.to_host(api.colnorm2(self.X)).sum())       # CGplvm.cpp:533-540
        L *= -0.5
        if not want_grad:
            return None, L
        G = api.covgrad_multi(invK, A)                                # sum_j updateCovGradient(j)
        gk = api.kern_grad(ks, self.X, G)                             # CKern::getGradParams, summed over j
        gk *= np.array([_gradfact(kind, x) for kind, x in zip(self.kinds, self._flat())])
        gX = api.to_host(api.kern_gradx(ks, self.X, G))               # CGplvm.cpp:573-604
        if self.regularise:
            gX = gX - self.X_host                                     # CGplvm.cpp:676-686
        return np.concatenate([gk, gX.reshape(-1, order="F")]), L

    def logLikelihood(self):
        return self.logLikelihoodGradient(want_grad=False)[1]
