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
    r, c = rank // pc, rank % pc
    groups = {}
    for rr in range(pr):          # every process creates every group, in the same order
        g = dist.new_group([rr * pc + cc for cc in range(pc)])
        if rr == r:
            groups[grid.AXIS_ROW] = (g, [rr * pc + cc for cc in range(pc)])
    for cc in range(pc):
        g = dist.new_group([rr * pc + cc for rr in range(pr)])
        if cc == c:
            groups[grid.AXIS_COL] = (g, [rr * pc + cc for rr in range(pr)])
    groups[grid.AXIS_WORLD] = (None, list(range(world)))
    on_gpu = flavour == "hip"
    if on_gpu:
        from gpc_amd import _lib
        lib = _lib.load()

    def as_tensor(ptr, count):
        if not on_gpu:
            return torch.from_numpy(np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_double)), shape=(count,)))
        t = torch.empty(count, dtype=torch.float64)
        _lib.check(lib.gpc_memcpy_d2h(t.data_ptr(), ptr, 8 * count, None))
        return t

    def put_back(ptr, t, count):
        if on_gpu:
            _lib.check(lib.gpc_memcpy_h2d(ptr, t.data_ptr(), 8 * count, None))

    def bcast(ptr, count, root, axis):
        g, ranks = groups[axis]
        t = as_tensor(ptr, count)
        dist.broadcast(t, ranks[root], group=g)
        put_back(ptr, t, count)

    def allreduce_sum(ptr, count, axis, on_device):
        g, ranks = groups[axis]
        if on_device:
            t = as_tensor(ptr, count)
            dist.all_reduce(t, group=g)
            put_back(ptr, t, count)
        else:
            t = torch.from_numpy(np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_double)), shape=(count,)))
            dist.all_reduce(t, group=g)

    def allreduce_min(v):
        t = torch.tensor([v], dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return int(t.item())

    binding = gc.host_binding() if flavour == "host" else None
    tr = grid.Transport(bcast, allreduce_sum, allreduce_min)
    g = grid.create_transport(rank, pr, pc, nb, tr, binding=binding)
    X, Y, Xs = gc.make_problem(N, D, d, Ns, 7)
    g.set_problem(gc.TERMS, X, Y, Xs)
    logdet, jit, info = g.update_k()
    out = {"logdet": logdet, "jitter": jit, "info": info, "ll": g.loglik(), "alpha": g.alpha()}
    if Ns:
        out["mu"], out["var"] = g.posterior()
    out["tiles"] = np.array(g.local_tiles(), dtype=object)
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), **out)
    g.destroy()
    dist.barrier()
    dist.destroy_process_group()
