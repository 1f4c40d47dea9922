#!/usr/bin/env python
"""Within-process interleaved A/B of the GEMM kernel variants on the Cholesky's shapes (run on the GPU box)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gpc_amd import api

def bench(fn, reps):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3

def main():
    variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,1,2").split(",")]
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    shapes = [("syrk", 16384, 512), ("syrk", 32768, 512), ("syrk", 8192, 512), ("syrk", 4096, 512),
              ("trap", 16384, 64), ("trsmgemm", 16384, 64)]
    torch.manual_seed(0)
    res = {}
    for kind, M, K in shapes:
        A = torch.randn((K if kind != "trap" else 64, M), dtype=torch.float64, device="cuda").t()   # M x K col-major
        if kind == "syrk":
            C = torch.randn((M, M), dtype=torch.float64, device="cuda").t()
            flops = M * (M + 1) * K
            def run():
                api.syrk(A, C, "L", "N", alpha=-1.0, beta=1.0)
        elif kind == "trap":
            nc = 448
            C = torch.randn((nc, M), dtype=torch.float64, device="cuda").t()
            flops = 2.0 * M * nc * 64 - nc * nc * 64
            lib = api.lib()
            def run():
                # M x nc x 64 update with both operands from A (full gemm entry; the internal trapezoid mask is not exposed)
                api.gemm(A, A[:nc, :], C, "N", "T", alpha=-1.0, beta=1.0)
            flops = 2.0 * M * nc * 64
        else:
            W = torch.randn((64, 64), dtype=torch.float64, device="cuda").t()
            flops = 2.0 * M * 64 * 64
            def run():
                api.gemm(A, W, A, "N", "T", alpha=1.0, beta=0.0)
        for r in range(rounds):
            for v in variants:
                api.check(api.lib().gpc_set_gemm_variant(v))
                run()
                t = bench(run, 3 if kind == "syrk" else 20)
                res.setdefault((kind, M, K, v), []).append(flops / t * 1e-12)
    for k, v in res.items():
        print("%-9s M=%6d K=%4d variant %d : TF/s median %.2f  min %.2f max %.2f" % (k[0], k[1], k[2], k[3], np.median(v), min(v), max(v)))

if __name__ == "__main__":
    main()
