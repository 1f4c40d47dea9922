"""Deterministic synthetic inputs for the parity tests and bench.py.

SplitMix64 counter stream -> Box-Muller normals, vectorised in numpy so that the same (seed, N, D) gives the same
arrays in this container (where the goldens are generated with the compiled reference) and on the GPU box.
X ~ N(0,1) iid, N x D;  y = sin(sum_k x_k / sqrt(D)) + 0.1 * N(0,1).
"""
import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)


def _splitmix64(idx, seed):
    with np.errstate(over="ignore"):
        z = (np.uint64(seed) + (idx + np.uint64(1)) * _GOLDEN)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def uniforms(n, seed, stream=0):
    """n doubles in (0,1), 53-bit."""
    idx = np.arange(n, dtype=np.uint64) + np.uint64(stream) * np.uint64(1 << 40)
    z = _splitmix64(idx, seed)
    return ((z >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def normals(n, seed, stream=0):
    m = (n + 1) // 2
    u1 = uniforms(m, seed, 2 * stream)
    u2 = uniforms(m, seed, 2 * stream + 1)
    r = np.sqrt(-2.0 * np.log(u1))
    out = np.empty(2 * m)
    out[0::2] = r * np.cos(2.0 * np.pi * u2)
    out[1::2] = r * np.sin(2.0 * np.pi * u2)
    return out[:n]


def make_xy(N, D, seed=1234):
    X = normals(N * D, seed, 0).reshape(D, N).T.copy()      # column d = contiguous block d (column-major fill)
    eps = normals(N, seed, 1)
    y = np.sin(X.sum(axis=1) / np.sqrt(D)) + 0.1 * eps
    return X, y.reshape(N, 1)


def make_xstar(Ns, D, seed=1234):
    return normals(Ns * D, seed, 2).reshape(D, Ns).T.copy()


# Kernel hyper-parameters of the BASELINE.json configs (natural space, all variances 1).
CONFIGS = {
    "cfg2": dict(N=8192, D=8, kern=[("rbf", [1.0, 1.0])]),
    "cfg3": dict(N=65536, D=32, kern=[("rbf", [2.0 / 32.0, 1.0]), ("white", [float(np.exp(-2.0))])]),
    "cfg4": dict(N=131072, D=16, kern=[("rbf", [1.0, 1.0])]),
}


def scaled_config(name, N):
    """Same kernel / D as a BASELINE config at an oracle-feasible N."""
    c = dict(CONFIGS[name])
    c["N"] = N
    return c
