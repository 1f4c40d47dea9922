"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol include/gpc_hip.h
declares, and the no-device behaviour is a loud error, not a fallback."""
import ctypes
import os
import re

import pytest

from gpc_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "gpc_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gpc_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert header_symbols() == sorted(_lib.SIGNATURES.keys())


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in header_symbols():
        assert hasattr(lib, name), "libgpc_hip.so does not export %s" % name


def test_no_device_is_a_loud_error():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = _lib.load()
    assert lib.gpc_version() >= 100
    n = ctypes.c_int(-1)
    assert lib.gpc_device_count(ctypes.byref(n)) == 0
    if n.value == 0:
        info = ctypes.c_int(0)
        rc = lib.gpc_potrf_f64(b"L", 4, None, 4, ctypes.byref(info), None)
        assert rc == _lib.GPC_ENODEV
        assert b"no CPU fallback" in lib.gpc_last_error()
        with pytest.raises(_lib.GpcError):
            _lib.check(rc)
