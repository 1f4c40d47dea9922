"""GPCB1 named-array container (Python twin of oracle/gpcb_io.h).

TEST INFRASTRUCTURE ONLY -- used by tests/, bench.py's cpu_baseline leg and the fixture generators to talk
to the oracle executables.  Nothing under gpc_amd/ imports this module.
"""
import struct
import numpy as np

MAGIC = b"GPCB1\n"


def write(path, arrays):
    """arrays: dict name -> array-like (scalars, 1-D (stored 1 x n) or 2-D); stored column-major fp64."""
    with open(path, "wb") as fp:
        fp.write(MAGIC)
        for name, val in arrays.items():
            a = np.asarray(val, dtype=np.float64)
            if a.ndim == 0:
                a = a.reshape(1, 1)
            elif a.ndim == 1:
                a = a.reshape(1, -1)
            nb = name.encode()
            fp.write(struct.pack("<i", len(nb)))
            fp.write(nb)
            fp.write(struct.pack("<qq", a.shape[0], a.shape[1]))
            fp.write(np.asfortranarray(a).tobytes(order="F"))


def read(path):
    out = {}
    with open(path, "rb") as fp:
        if fp.read(6) != MAGIC:
            raise ValueError("not a GPCB1 file: %s" % path)
        while True:
            h = fp.read(4)
            if len(h) < 4:
                break
            (nl,) = struct.unpack("<i", h)
            name = fp.read(nl).decode()
            rows, cols = struct.unpack("<qq", fp.read(16))
            buf = fp.read(8 * rows * cols)
            out[name] = np.frombuffer(buf, dtype=np.float64).reshape((rows, cols), order="F").copy()
    return out
