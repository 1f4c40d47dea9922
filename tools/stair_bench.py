"""Trailing-update launch in isolation: compact triangle (single GPU) vs the 2-D staircase a grid rank runs.
usage: python tools/stair_bench.py m nb [reps]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpc_amd import _lib  # noqa: E402

lib = _lib.load()
f = lib.gpc_bench_update
f.restype = ctypes.c_int
f.argtypes = [ctypes.c_int, ctypes.c_int64, ctypes.c_int64] + [ctypes.c_int] * 5 + [ctypes.POINTER(ctypes.c_double)] * 2
m, nb = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5


def run(mode, mm, pr, pc, r, c):
    ms, fl = ctypes.c_double(0), ctypes.c_double(0)
    _lib.check(f(mode, mm, nb, pr, pc, r, c, reps, ctypes.byref(ms), ctypes.byref(fl)))
    return ms.value, fl.value


for label, mode, mm, pr, pc, r, c in [("compact triangle", 1, m, 1, 1, 0, 0), ("in place: compact triangle", 6, m, 1, 1, 0, 0),
                                      ("in place: 1x1 staircase + 16 extra rows", 7, m, 1, 1, 0, 0), ("staircase 1x1", 5, m, 1, 1, 0, 0),
                                      ("staircase 2x4 (0,0) of 2m x 4m... m_glob=%d" % (2 * m), 5, 2 * m, 2, 4, 0, 0),
                                      ("staircase 2x4 (1,3)", 5, 2 * m, 2, 4, 1, 3),
                                      ("staircase 2x2 (1,0)", 5, 2 * m, 2, 2, 1, 0)]:
    ms, fl = run(mode, mm, pr, pc, r, c)
    print("%-45s %8.3f ms  %6.1f TF" % (label, ms, fl / ms * 1e-9))
