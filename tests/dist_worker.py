"""Worker for the multi-process block-cyclic tests (spawned by tests/test_dist_cholesky.py): one rank of a
torch.distributed job, either the CPU/gloo + numpy-ops flavour or the GPU flavour (real HIP ops; the ranks may share
one GPU, in which case the collectives run over gloo, staged through the host)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def make_problem(N, D, d, Ns, seed):
    rng = np.random.RandomState(seed)
    X = rng.randn(N, D)
    y = np.sin(X[:, :1] * np.arange(1, d + 1)[None, :]) + 0.05 * rng.randn(N, d)
    Xs = rng.randn(Ns, D)
    return X, y, Xs


def run(rank, world, port, flavour, N, D, d, Ns, nb, terms, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sync = flavour.endswith("-sync")               # the serialised fall-back mode of gpc_amd.dist
    if sync:
        flavour = flavour[:-5]
        os.environ["GPC_DIST_SYNC"] = "1"
    if flavour == "hip-rccl":                      # real RCCL; one rank per GPU (a 1-GPU box can only run world = 1)
        torch.cuda.set_device(rank)
        os.environ["GPC_DIST_FORCE_COMM"] = "1"
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank),
                                pg_options=dist.ProcessGroupNCCL.Options(is_high_priority_stream=True))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gpc_amd import dist as gdist
        if flavour == "numpy":
            from dist_numpy_ops import NumpyOps
            ops = NumpyOps()
        elif flavour == "hip-rccl":
            ops = gdist.HipOps()
        else:
            torch.cuda.set_device(0)
            ops = gdist.HipOps()
        X, y, Xs = make_problem(N, D, d, Ns, 7)
        g = gdist.DistGp(terms, X, y, Xs if Ns else None, nb=nb, ops=ops)
        logdet = g.update_k()
        ll = g.log_likelihood()
        al = g.alpha()
        res = {"logdet": logdet, "ll": ll, "alpha": al.cpu().numpy().copy(), "ncols": g.ncols, "jitter": g.jitter}
        if Ns:
            mu, var = g.posterior(al)
            res["mu"] = mu.cpu().numpy().copy()
            res["var"] = var.cpu().numpy().copy()
        res["L"] = g.gather_factor() if flavour != "hip-rccl" else np.tril(g.A[:N, :N].cpu().numpy())
        np.savez(os.path.join(outdir, "rank%d.npz" % rank), **res)
    finally:
        dist.destroy_process_group()


def singular_inputs():
    rng = np.random.RandomState(3)
    X = rng.randn(300, 2)
    X[150:] = X[:150]            # every point twice: K is exactly singular
    return X


def run_singular(rank, world, port, terms, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gpc_amd import dist as gdist
        from dist_numpy_ops import NumpyOps
        g = gdist.DistGp(terms, singular_inputs(), nb=128, ops=NumpyOps())
        logdet = g.update_k()
        np.savez(os.path.join(outdir, "sing%d.npz" % rank), jitter=g.jitter, logdet=logdet)
    finally:
        dist.destroy_process_group()
