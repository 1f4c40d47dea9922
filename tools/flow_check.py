"""panel_flow.hip against the launch chain: same factor (to rounding), time per factor.  usage: flow_check.py N [reps]"""
import os, sys, time, subprocess
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 2 and sys.argv[2] == "child":
    import torch
    from gpc_amd import api, synth
    N = int(sys.argv[1])
    X, _ = synth.make_xy(N, 8, 1234)
    ks = api.kspec([("rbf", [1.0, 1.0]), ("white", [0.01])])
    Xd = api.from_host(X)
    K = api.empty(N, N)
    for _ in range(2):
        L, ld, jit, info = api.gp_update_k(ks, Xd, K)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        L, ld, jit, info = api.gp_update_k(ks, Xd, K)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    np.save(sys.argv[3], np.tril(api.to_host(L)))
    if os.environ.get("GPC_PANEL_FLOW_TRACE") and os.environ.get("GPC_PANEL_FLOW") == "1":
        import ctypes
        from gpc_amd import _lib
        lib = _lib.load()
        buf = (ctypes.c_longlong * (64 * 64 * 4))()
        lib.gpc_debug_panel_flow_trace(buf, ctypes.c_int64(64 * 64 * 4))
        tr = np.array(buf, dtype=np.int64).reshape(64, 64, 4)
        nb = min(16, (N + 63) // 64)
        t0 = tr[0, 0, 0]
        us = lambda x: (x - t0) / 100.0
        if os.environ.get("GPC_PANEL_FLOW_TRACE") == "2":
            f = tr.reshape(-1)[63 * 256: 63 * 256 + 64].reshape(8, 8)
            print("chol of block (0,0), per column group (8 groups of 8 or 4 of 16) (us): stage+sync / pivots / update / publish")
            for k in range(8):
                if f[k, 0] == 0:
                    continue
                print("   blk %d: %.2f %.2f %.2f %.2f" % (k, (f[k,1]-f[k,0])/100., (f[k,2]-f[k,1])/100., (f[k,3]-f[k,2])/100., (f[k,4]-f[k,3])/100.))
        if os.environ.get("GPC_PANEL_FLOW_TRACE") == "2":
            f = tr.reshape(-1)[61 * 256: 61 * 256 + 64].reshape(8, 8)
            if f[0, 0] != 0:
                print("chol of block (0,0), wave 0 (an updating / publishing wave) per group (us): rest update, publish, wait B, tile column + stage, wait A")
                for k in range(8):
                    print("   blk %d: %.2f %.2f %.2f %.2f %.2f" % ((k,) + tuple(max(f[k, i + 1] - f[k, i], 0) / 100. for i in range(5))))
            f = tr.reshape(-1)[62 * 256: 62 * 256 + 32].reshape(4, 8)
            print("solve of block (1,0), per 16-column group (us since chol start): start | L there, staged+sync, triangle, trailing, publish")
            for k in range(4):
                print("   blk %d: %.2f | %.2f %.2f %.2f %.2f %.2f" % ((k, us(f[k, 0])) + tuple((f[k, i + 1] - f[k, i]) / 100. for i in range(5))))
        print("trace (us since start): block (b, c): start / products done / S ready / end")
        for c in range(nb):
            for b in (c, c + 1, nb - 1, min(63, (N + 63) // 64 - 1)):
                if b < 64 and b >= c and b * 64 < N:
                    print("  b=%2d c=%2d  %8.1f %8.1f %8.1f %8.1f" % ((b, c) + tuple(us(x) for x in tr[b, c])))
    print("flow=%s N=%d info=%d logdet=%.15g  %.3f ms per factor (%.1f TF)" % (os.environ.get("GPC_PANEL_FLOW", "0"), N, info, ld, dt * 1e3, N ** 3 / 3.0 / dt * 1e-12))
else:
    N = sys.argv[1]
    outs = []
    for flow in ("0", "1"):
        f = "/tmp/flow_%s.npy" % flow
        r = subprocess.run([sys.executable, __file__, N, "child", f], env=dict(os.environ, GPC_PANEL_FLOW=flow), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
        print(r.stdout.decode().strip().splitlines()[-1] if r.stdout else "no output", "rc", r.returncode)
        outs.append(np.load(f) if os.path.exists(f) else None)
    if outs[0] is not None and outs[1] is not None:
        print("max |L_flow - L_chain| / max|L| = %.3e" % (np.abs(outs[0] - outs[1]).max() / np.abs(outs[0]).max()))
