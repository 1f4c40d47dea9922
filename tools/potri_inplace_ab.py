#!/usr/bin/env python
"""dpotri: the in-place form (round 4) against the N x N-workspace form, size by size (run on the GPU box).
usage: potri_inplace_ab.py N [N ...]   env GPC_POTRI_LAUUM_NB = block width of the in-place second phase"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpc_amd import api

def bench(fn, reps=2):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

for N in [int(a) for a in sys.argv[1:]] or [16384]:
    D = 8
    X = torch.randn((D, N), dtype=torch.float64, device="cuda").t()
    ks = api.kspec([("rbf", [1.0, 1.0]), ("white", [0.1])])
    L = api.empty(N, N); W = api.empty(N, N)
    api.gram_sym(ks, X, L)
    assert api.potrf(L, "L") == 0
    t_copy = bench(lambda: W.copy_(L))
    out = {}
    for name, minn in (("inplace", "2048"), ("workspace", str(1 << 40))):
        os.environ["GPC_POTRI_INPLACE_MINN"] = minn
        def f():
            W.copy_(L); api.potri(W, "L")
        try:
            f()
            t = bench(f, 3 if N <= 16384 else 1) - t_copy
            out[name] = (t, W[:: max(N // 64, 1), :: max(N // 64, 1)].clone())
        except Exception as e:
            out[name] = (float("nan"), None)
            print("N=%d %s failed: %s" % (N, name, str(e)[:120]))
        api.lib().gpc_workspace_release()
    d = float((out["inplace"][1] - out["workspace"][1]).abs().max() / out["workspace"][1].abs().max()) if out["inplace"][1] is not None and out["workspace"][1] is not None else float("nan")
    print("N=%6d  in place %9.2f ms (%.1f TF)   workspace %9.2f ms (%.1f TF)   sampled rel diff %.1e" % (
        N, out["inplace"][0], 2 * N ** 3 / 3 / out["inplace"][0] * 1e-9, out["workspace"][0], 2 * N ** 3 / 3 / out["workspace"][0] * 1e-9, d), flush=True)
    del L, W, X
    torch.cuda.empty_cache()
