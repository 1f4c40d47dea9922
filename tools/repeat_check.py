"""The same factorisation and inverse twice: bit for bit the same?  (tall panels through the tile's inverse, k-limited products,
dataflow right-sided solves.)  usage: python tools/repeat_check.py N [N ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from gpc_amd import api, synth  # noqa: E402

for N in [int(a) for a in sys.argv[1:]]:
    X, _ = synth.make_xy(N, 8, 5)
    ks = api.kspec([("rbf", [0.25, 1.0]), ("white", [0.05])])
    Xd = api.from_host(X)
    res = []
    for rep in range(2):
        K = api.empty(N, N)
        L, ld, jit, info = api.gp_update_k(ks, Xd, K)
        Lc = torch.tril(L).clone()
        api.potri(L, "L")
        res.append((ld, Lc, L.clone()))
        del K, L
    same_l = torch.equal(res[0][1], res[1][1])
    same_i = torch.equal(res[0][2], res[1][2])
    print("N=%d logdet %r / %r  factor identical: %s  inverse identical: %s" % (N, res[0][0], res[1][0], same_l, same_i))
    assert res[0][0] == res[1][0] and same_l and same_i
    del res
    torch.cuda.empty_cache()
