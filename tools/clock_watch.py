"""Shader clock and power while (a) the trailing-update kernel, (b) the pure fp64-MFMA probe loop runs for a few seconds each:
sampled from rocm-smi in a side thread.  usage: python tools/clock_watch.py [M] [K]"""
import ctypes
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from gpc_amd import api  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 49152
K = int(sys.argv[2]) if len(sys.argv) > 2 else 1536
samples, stop = [], [False]


def sampler():
    while not stop[0]:
        try:
            out = subprocess.run(["rocm-smi", "-d", "0", "--showclocks", "--showpower"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                                 timeout=10).stdout.decode()
            sclk = [ln for ln in out.splitlines() if "sclk" in ln.lower()]
            pw = [ln for ln in out.splitlines() if "power" in ln.lower() and "(w)" in ln.lower()]
            samples.append((time.time(), sclk[0].split(":")[-1].strip() if sclk else "?", pw[0].split(":")[-1].strip() if pw else "?"))
        except Exception as e:  # noqa: BLE001
            samples.append((time.time(), "err %s" % e, ""))
        time.sleep(0.05)


A = torch.randn((K, M), dtype=torch.float64, device="cuda").t()
C = torch.randn((M, M), dtype=torch.float64, device="cuda").t()
api.syrk(A, C, "L", "N", alpha=-1e-9, beta=1.0)
torch.cuda.synchronize()
th = threading.Thread(target=sampler)
th.start()
time.sleep(1.0)
marks = [("idle", time.time())]
t0 = time.time()
n = 0
while time.time() - t0 < 6.0:
    for _ in range(4):
        api.syrk(A, C, "L", "N", alpha=-1e-9, beta=1.0)
        n += 1
    torch.cuda.synchronize()
dt = time.time() - t0
marks.append(("syrk M=%d K=%d: %.1f TFLOP/s" % (M, K, n * M * (M + 1.0) * K / dt * 1e-12), time.time()))
time.sleep(1.0)
marks.append(("idle", time.time()))
t0 = time.time()
res = None
while time.time() - t0 < 6.0:
    tf, cyc, ghz = ctypes.c_double(0), ctypes.c_double(0), ctypes.c_double(0)
    api.check(api.lib().gpc_probe_mfma_f64(ctypes.byref(tf), ctypes.byref(cyc), ctypes.byref(ghz), api.stream()))
    res = "%.1f TFLOP/s" % tf.value
marks.append(("mfma probe loop: %r" % (res,), time.time()))
time.sleep(0.5)
stop[0] = True
th.join()
lo = samples[0][0]
for name, tend in marks:
    seg = [s for s in samples if lo <= s[0] < tend]
    print("== %s: %d samples" % (name, len(seg)))
    print("   sclk: %s" % ", ".join(s[1] for s in seg[:: max(1, len(seg) // 12)]))
    print("   power: %s" % ", ".join(s[2] for s in seg[:: max(1, len(seg) // 12)]))
    lo = tend
