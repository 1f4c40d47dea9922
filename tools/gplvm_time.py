#!/usr/bin/env python
"""Wall time of `gplvm learn` on the oil data (BASELINE config 5), new kernels against the stepped ones (run on the GPU box)."""
import os, subprocess, sys, time, re
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
exe, data = os.path.join(R, "gpc_amd", "host", "gplvm"), os.path.join(R, "tests", "golden", "oilTrain.svml")
iters = sys.argv[1] if len(sys.argv) > 1 else "30"
OLD = dict(GPC_TRSV_FLOW="0", GPC_TRSM_CHAIN="0", GPC_PANEL_STEP="0", GPC_POTF2="0", GPC_NB="512")
for label, extra in (("warm-up", {}), ("new", {}), ("old", OLD)):
    t0 = time.time()
    r = subprocess.run([exe, "-v", "3", "-s", "1", "learn", "-#", iters, data, "/tmp/oil_%s.model" % label],
                       env=dict(os.environ, **extra), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    dt = time.time() - t0
    out = r.stdout.decode()
    ev = re.findall(r"Objective evaluations: (\d+)\s+gradient evaluations: (\d+)", out)
    print(label, "rc", r.returncode, "wall %.2f s" % dt, "iterations", len(re.findall(r"^Iteration", out, flags=re.M)), "evals", ev[-1:] )
