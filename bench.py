#!/usr/bin/env python
"""bench.py -- N x N RBF Gram build + Cholesky factors/sec on MI355X (BASELINE.json's metric).

One "step" = one CGp::updateK()-equivalent (FTC): Gram build of the config's kernel from X resident in HBM, blocked
Cholesky in place, log-determinant (gpc_gp_update_k_f64).  Default workload = BASELINE config 3 (N = 65 536, D = 32,
rbf + white, the configuration the north-star target is quoted on; K is 34.4 GB and fits one 288 GB GPU).
`--workload cfg2` runs config 2 (N = 8 192, D = 8, rbf).

N > 1 GPUs (launched by torch.distributed.run, one process per GPU): ONE factorisation of the same workload spread
over the ranks by gpc_amd/dist.py (1-D block-cyclic column panels, panel broadcasts over RCCL/xGMI, look-ahead), so
the total work is fixed ("scaling": "strong") and `value` is still whole-job factors/s.  GPC_BENCH_REPLICAS=1 runs
independent replicas instead (weak scaling, no collective).

Prints ONE JSON line on rank 0.  The `roofline` object is measured live with HIP events bracketing the dominant
kernel (the trailing SYRK update on fp64 MFMA) on the stream it is launched on; `cpu_baseline` times the compiled
reference (oracle/_ref, kind "reference") or, when that binary or MKL is absent, the C restatement (kind "port")
on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6      # 256 CU x 4 SIMD x 2.4 GHz x 32 flop/clk (AMD MI355X fp64 matrix)
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: 8 TB/s HBM3E


def cpu_baseline(cfg, sample_n, seed):
    """Reference semantics of one updateK (scalar Gram loop + jitChol) on a bounded sample, on the host cores."""
    from gpc_amd import synth
    from oracle import refrun, portrun
    X, _ = synth.make_xy(sample_n, cfg["D"], seed)
    cores = min(os.cpu_count() or 1, 64)     # BLAS threads; more than 64 only slow a factorisation of this size down
    t0 = time.time()
    if refrun.have_ref():
        arrays = dict(refrun.kern_arrays(cfg["kern"]))
        arrays.update({"X": X, "reps": 1.0})
        r = refrun.run_ref("time", arrays, threads=cores)
        kind, used = "reference", cores
    elif refrun.have_port():
        sample_n = min(sample_n, 2048)
        X = X[:sample_n]
        r = portrun.time_update_k(cfg["kern"], X, reps=1)
        kind, used = "port", 1
    else:
        return None
    tg, tc = float(r["t_gram"][0, 0]), float(r["t_chol"][0, 0])
    return {"value": 1.0 / (tg + tc), "unit": "factors/s at the sample size", "cores": used, "kind": kind,
            "sample": "N=%d D=%d same kernel, one CGp::updateK (scalar Gram loop %.2f s + jitChol %.2f s; "
                      "BLAS threads = cores for the reference, the Gram loop is single-threaded); wall %.1f s"
                      % (sample_n, cfg["D"], tg, tc, time.time() - t0)}


def pmc_traffic(workload, single_gpu_path):
    """HBM bytes per trailing-update launch from the committed PMC passes of THIS command (tools/pmc_bench_traffic.sh ->
    profiles/rNN_pmc_bench_traffic.json); None when no such profile exists for the workload being run."""
    import glob
    if workload != "cfg3" or not single_gpu_path:
        return None, None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_bench_traffic.json")))
    if not files:
        return None, None
    try:
        with open(files[-1]) as f:
            j = json.load(f)
        return float(j["hbm_bytes_per_launch"]), os.path.relpath(files[-1], ROOT)
    except (OSError, ValueError, KeyError):
        return None, None


def jobs_flops(world, replicas):
    return world if replicas else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=os.environ.get("GPC_BENCH_WORKLOAD", "cfg3"))
    ap.add_argument("--n", type=int, default=0, help="override N (debug)")
    ap.add_argument("--cpu-sample-n", type=int, default=4096)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    if not os.path.exists(os.path.join(ROOT, "gpc_amd", "lib", "libgpc_hip.so")) and \
            int(os.environ.get("LOCAL_RANK", "0")) == 0 and int(os.environ.get("WORLD_SIZE", "1")) == 1:
        import __graft_entry__           # a checkout without the built library: build it (single-process runs only)
        __graft_entry__.build()
    from gpc_amd import api, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)   # panel broadcasts overtake the SYRK
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), pg_options=opts)
    api.lib()
    replicas = world > 1 and os.environ.get("GPC_BENCH_REPLICAS", "0") == "1"
    # GPC_BENCH_DIST=1: drive the block-cyclic code path on ONE GPU too (P = 1, no collective) to price its overhead
    distributed = (world > 1 and not replicas) or os.environ.get("GPC_BENCH_DIST", "0") == "1"

    cfg = dict(synth.CONFIGS[args.workload])
    if args.n:
        cfg["N"] = args.n
    N, D = cfg["N"], cfg["D"]
    X, _ = synth.make_xy(N, D, seed=1234 + (rank if replicas else 0))
    dist_fallback = False
    if distributed:
        from gpc_amd import dist as gdist
        dist_mode = "overlapped"
        if world > 1 and os.environ.get("GPC_BENCH_SELFCHECK", "1") == "1":
            # Start-up self-check (untimed): the block-cyclic factorisation of a small problem must give the log-determinant
            # a single-GPU factorisation gives, on every rank.  The overlapped mode (second stream, asynchronous RCCL
            # broadcasts ordered by events) is checked first; if it disagrees the run falls back to the serialised mode.
            Xc, _ = synth.make_xy(8192, D, seed=99)
            _, ref_ld, _, info0 = api.gp_update_k(api.kspec(cfg["kern"]), api.from_host(Xc))
            for mode in ("overlapped", "serialised"):
                ok = 0.0
                try:
                    gc = gdist.DistGp(cfg["kern"], Xc, sync=(mode == "serialised"))
                    ld = gc.update_k()
                    ok = 1.0 if (info0 == 0 and abs(ld - ref_ld) <= 1e-8 * abs(ref_ld)) else 0.0
                    del gc
                except Exception as e:     # noqa: BLE001 -- any failure of the check means "do not use this mode"
                    sys.stderr.write("rank %d: distributed self-check (%s) failed: %r\n" % (rank, mode, e))
                flag = torch.tensor([ok], dtype=torch.float64, device="cuda")
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                if float(flag.item()) == 1.0:
                    dist_mode = mode
                    break
            else:
                # neither mode reproduces the single-GPU factor on this node: report what the GPUs do independently
                # rather than nothing (every rank reaches this branch together: the flags were all-reduced)
                sys.stderr.write("rank %d: block-cyclic factorisation failed its self-check in both modes; "
                                 "running %d independent replicas instead\n" % (rank, world))
                distributed, replicas, dist_fallback = False, True, True
                X, _ = synth.make_xy(N, D, seed=1234 + rank)
    if distributed:
        g = gdist.DistGp(cfg["kern"], X, sync=(dist_mode == "serialised"))

        def step():
            return g.update_k()
    else:
        Xd = api.from_host(X)
        ks = api.kspec(cfg["kern"])
        # leading dimension of K: N by default (like CMatrix); GPC_BENCH_LDPAD adds rows of padding to move the column
        # stride off a power of two (HBM channel interleaving), an experiment knob
        ldpad = int(os.environ.get("GPC_BENCH_LDPAD", "0"))
        K = api.empty(N + ldpad, N)[:N, :] if ldpad > 0 else api.empty(N, N)

        def step():
            _, logdet, jit, info = api.gp_update_k(ks, Xd, K)
            assert info == 0, "factorisation failed (info=%d)" % info
            return logdet

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    api.check(api.lib().gpc_profile_enable(1))
    import ctypes
    for kind in (0, 1):
        api.check(api.lib().gpc_profile_read(kind, None, None, None, 1))
    sync()
    t0 = time.perf_counter()
    logdet = 0.0
    for _ in range(args.steps):
        logdet = step()
    sync()
    dt = time.perf_counter() - t0
    api.check(api.lib().gpc_profile_enable(0))
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    def prof(kind):
        n, ms, w = ctypes.c_int64(0), ctypes.c_double(0.0), ctypes.c_double(0.0)
        api.check(api.lib().gpc_profile_read(kind, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(w), 1))
        return n.value, ms.value, w.value

    syrk_n, syrk_ms, syrk_flops = prof(0)
    # algorithmic HBM bytes of the trailing updates of one factor: each reads its panel rows once (8*m*NB) and reads +
    # writes the lower triangle it updates (2 * 8 * m(m+1)/2); summed over the panels
    syrk_bytes, k0 = 0.0, 0
    fixed_nb = int(os.environ.get("GPC_NB", "0"))
    while k0 < N:
        rem = N - k0
        nbp = fixed_nb if fixed_nb >= 64 else 1024   # potrf.hip panel_width()
        nbp = min(nbp, rem)
        m = rem - nbp
        if m > 0:
            syrk_bytes += 8.0 * m * nbp + 8.0 * m * (m + 1)
        k0 += nbp
    syrk_bytes *= args.steps
    gram_n, gram_ms, gram_bytes = prof(1)

    phases = None
    if rank == 0 and not distributed and not replicas and os.environ.get("GPC_BENCH_PHASES", "1") == "1":
        # The other phases of one likelihood / gradient evaluation on the factor just computed, timed with events on
        # torch's current stream (the stream every C-ABI call above ran on).  NOT part of `value`.
        def timed(fn):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1), r
        _, yv = synth.make_xy(N, D, seed=1234)
        yd = api.from_host(yv)
        al = api.empty(N, 1)
        t_alpha, _ = timed(lambda: api.gp_alpha(K, yd, out=al))            # CGp::updateAlpha: two triangular solves
        t_ll, ll = timed(lambda: api.gp_loglik(yd, al, logdet))
        phases = {"alpha_2trsv_ms": t_alpha, "alpha_algorithmic_GBs": 8.0 * N * N / (t_alpha * 1e-3) * 1e-9,
                  "loglik_ms": t_ll, "loglik": ll}
        try:
            inv = K.clone()
            api.potri(inv, "L")                                            # first call allocates the N x N workspace
            inv.copy_(K)
            t_potri, _ = timed(lambda: api.potri(inv, "L"))                # CMatrix::pdinv for the gradient
            phases.update({"potri_ms": t_potri, "potri_tflops_at_2N3_over_3": 2.0 * N ** 3 / 3.0 / (t_potri * 1e-3) * 1e-12})
            # CKern::getGradParams over a symmetric N x N covGrad (the inverse stands in for it: same bytes, same work)
            t_kg, _ = timed(lambda: api.kern_grad(ks, Xd, inv))
            phases.update({"kern_grad_ms": t_kg, "kern_grad_GBs_of_4N2_bytes": 4.0 * N * N / (t_kg * 1e-3) * 1e-9})
            del inv
        except (RuntimeError, api._lib.GpcError) as e:                      # not enough HBM for the two extra N x N buffers
            phases["potri_ms"] = None
            phases["potri_skipped"] = str(e)[:100]

    if rank == 0:
        probe, pcyc, pclk = ctypes.c_double(0.0), ctypes.c_double(0.0), ctypes.c_double(0.0)
        api.check(api.lib().gpc_probe_mfma_f64(ctypes.byref(probe), ctypes.byref(pcyc), ctypes.byref(pclk),
                                               api.stream()))
        achieved = syrk_flops / (syrk_ms * 1e-3) * 1e-12 if syrk_ms > 0 else 0.0
        potrf_flops = N ** 3 / 3.0
        traffic, traffic_src = pmc_traffic(args.workload if not args.n else "custom", not distributed)
        roof = {"bound": "mfma", "kernel": "gemm_nt_fast_kernel<4, 1> (trailing SYRK launches U1+U2 of %s)"
                                           % ("gpc_syrk_blockcyclic_f64, rank 0" if distributed else "gpc_potrf_f64"),
                "achieved": achieved, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / FP64_MFMA_PEAK_TFLOPS, "traffic": traffic,
                "traffic_unit": "HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024, rocprofv3 --pmc passes of "
                                "this command", "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": syrk_bytes / max(1, syrk_n),
                "launches_per_step": syrk_n / max(1, args.steps),
                "avg_launch_ms": syrk_ms / max(1, syrk_n),
                "algorithmic_flops_per_launch": syrk_flops / max(1, syrk_n),
                "mfma_f64_probe_tflops": probe.value, "mfma_f64_probe_cycles_per_mfma": pcyc.value,
                "mfma_f64_probe_clock_ghz": pclk.value,
                "whole_factor_tflops": jobs_flops(world, replicas) * potrf_flops * args.steps / dt * 1e-12,
                "gram": {"bound": "hbm", "achieved": gram_bytes / (gram_ms * 1e-3) * 1e-9 if gram_ms > 0 else 0.0,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "avg_launch_ms": gram_ms / max(1, gram_n)}}
        roof["gram"]["frac"] = roof["gram"]["achieved"] / HBM_PEAK_GBS
        jobs = world if replicas else 1
        out = {"metric": "N x N RBF Gram build + Cholesky factors/sec", "value": jobs * args.steps / dt,
               "unit": "factors/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
               "scaling": "strong" if distributed else "weak",
               "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": "%s: N=%d D=%d kernel=%s, one CGp::updateK (Gram + dpotrf + logdet) per step"
                                      % (args.workload, N, D, "+".join(t for t, _ in cfg["kern"])),
                          "parallelism": "1 GPU" if world == 1 else
                          ("%d independent replicas%s" % (world, " (the block-cyclic factorisation failed its start-up "
                                                          "self-check on this node)" if dist_fallback else "")
                           if replicas else
                           "1-D block-cyclic column panels over %d GPUs (nb=%d), RCCL panel broadcast, %s"
                           % (world, g.nb, "look-ahead 1, slab-pipelined" if dist_mode == "overlapped" else
                              "serialised collectives (the overlapped mode failed the start-up self-check)")),
                          "logdet": logdet},
               "roofline": roof}
        if phases is not None:
            phases["gram_ms"] = gram_ms / max(1, gram_n)
            phases["potrf_logdet_ms"] = dt / args.steps * 1e3 - phases["gram_ms"]
            out["phases"] = phases
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, min(args.cpu_sample_n, N), 1234)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
