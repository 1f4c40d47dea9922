import ctypes, torch, numpy as np
lib = ctypes.CDLL("/root/repo/tools/libgpc_dbg.so")
dbg = torch.zeros(8, dtype=torch.int64, device="cuda")
lib.gpc_dbg_set(ctypes.c_void_p(dbg.data_ptr()))
A = (torch.eye(64, dtype=torch.float64, device="cuda") * 64 + 1.0).t().contiguous().t()
info = ctypes.c_int(0)
lib.gpc_potrf_f64.argtypes = [ctypes.c_char, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_int), ctypes.c_void_p]
for i in range(3):
    A.copy_(torch.eye(64, dtype=torch.float64, device="cuda") * 64 + 1.0)
    lib.gpc_potrf_f64(b"L", 64, ctypes.c_void_p(A.data_ptr()), 64, ctypes.byref(info), None)
    torch.cuda.synchronize()
    print("ticks: load %d factor %d scale %d inverse %d store %d" % tuple(dbg[:5].tolist()))
