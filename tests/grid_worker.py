"""One rank of a multi-process grid job whose exchange is torch.distributed (gloo) behind gpc_grid_create_transport.
Used by tests/test_grid_cpu.py (host stand-in) and tests/test_grid_gpu.py (HIP library, ranks sharing the one GPU)."""
import ctypes
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def run(rank, world, port, pr, pc, nb, N, D, d, Ns, outdir, flavour):
    import torch
    import torch.distributed as dist
    import grid_common as gc
    from gpc_amd import grid
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    on_gpu = flavour == "hip"
    tr = grid.torch_transport(rank, pr, pc, on_device=on_gpu)
    binding = gc.host_binding() if flavour == "host" else None
    g = grid.create_transport(rank, pr, pc, nb, tr, binding=binding)
    X, Y, Xs = gc.make_problem(N, D, d, Ns, 7)
    g.set_problem(gc.TERMS, X, Y, Xs)
    logdet, jit, info = g.update_k()
    out = {"logdet": logdet, "jitter": jit, "info": info, "ll": g.loglik(), "alpha": g.alpha()}
    if Ns:
        out["mu"], out["var"] = g.posterior()
    out["tiles"] = np.array(g.local_tiles(), dtype=object)
    out["grad"] = g.gradient(len(gc.expected_gradient(gc.TERMS, X[:4], Y[:4])))     # (the sums' count; the values come from all of X)
    out["held"] = g.stats()["bytes_held"]
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), **out)
    g.destroy()
    dist.barrier()
    dist.destroy_process_group()
