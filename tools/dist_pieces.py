#!/usr/bin/env python
"""Time the per-rank pieces of the block-cyclic factorisation on ONE GPU (no collectives): the tall-panel factor, the
pack copy and the staircase update of a rank's share, for a P-rank job at N.  Feeds the scaling model in DESIGN.md."""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpc_amd import api  # noqa: E402


def timeit(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=65536)
    ap.add_argument("--p", type=int, default=8)
    ap.add_argument("--nb", type=int, default=512)
    ap.add_argument("--rank", type=int, default=3)
    args = ap.parse_args()
    N, P, nb, r = args.n, args.p, args.nb, args.rank
    T = N // nb
    mine = list(range(r, T, P))
    ncols = len(mine) * nb
    A = api.empty(N, ncols)
    A.fill_(0.0)
    out = []
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    for k in [0, T // 4, T // 2, 3 * T // 4]:
        M = N - k * nb
        # a well-conditioned tall panel: identity-ish diagonal block on top of small random rows
        pan = api.empty(M, nb)
        pan.copy_(torch.rand((M, nb), dtype=torch.float64, device="cuda") * 1e-3)
        pan[:nb, :] += torch.eye(nb, dtype=torch.float64, device="cuda") * 4.0
        keep = pan.clone()
        def fact():
            pan.copy_(keep)
            api.potrf_panel(pan, k * nb, info)
        t_copy = timeit(lambda: pan.copy_(keep))
        t_fact = timeit(fact) - t_copy
        def fact_slabs(sb=128):
            pan.copy_(keep)
            for c0 in range(0, nb, sb):
                api.potrf_panel(pan[c0:, c0:c0 + sb], k * nb + c0, info)
                if c0 + sb < nb:
                    r0 = k * nb + c0 + sb
                    api.syrk_blockcyclic(pan[c0 + sb:, c0:c0 + sb], pan[c0 + sb:, c0 + sb:nb], r0, r0 // sb, 1, sb)
        t_slab = timeit(fact_slabs) - t_copy
        flat = torch.empty(M * nb, dtype=torch.float64, device="cuda")
        view = flat.view(nb, M).t()
        src = A[k * nb:, 0:nb]
        t_pack = timeit(lambda: view.copy_(src))
        row0 = (k + 1) * nb
        l0 = 0 if k < r else (k - r) // P + 1
        Pk = view[nb:, :]
        C = A[row0:, l0 * nb:ncols]
        def upd():
            api.syrk_blockcyclic(Pk, C, row0, r + l0 * P, P, nb)
        t_upd = timeit(upd)
        entries = 0.0
        for l in range(l0, len(mine)):
            g0 = mine[l] * nb
            entries += (N - g0) * nb - 0.5 * nb * (nb - 1)
        assert int(info.item()) == 0
        out.append({"k": k, "M": M, "panel_factor_ms": t_fact, "panel_factor_128slabs_ms": t_slab, "pack_ms": t_pack, "update_ms": t_upd,
                    "update_tflops": 2.0 * nb * entries / (t_upd * 1e-3) * 1e-12,
                    "panel_bytes_MB": M * nb * 8 / 1e6})
    print(json.dumps({"N": N, "P": P, "nb": nb, "rank": r, "pieces": out}))


if __name__ == "__main__":
    main()
