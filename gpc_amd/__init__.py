"""gpc_amd -- MI355X-native exact-GP hot path of GPc (Gram build -> Cholesky -> solves / log-det).

The product is the C-ABI shared library gpc_amd/lib/libgpc_hip.so (include/gpc_hip.h) plus the C++ host classes under
gpc_amd/host/ that keep GPc's CMatrix / CKern / CGp surface.  This Python package is plumbing for tests and bench.py:
`gpc_amd.api` forwards torch device tensors to the C-ABI, `gpc_amd.gp.CGp` mirrors the reference's FTC CGp methods on
top of it.  Importing the package does not load the library; the first call does, and it raises if the library is
missing (there is no CPU fallback anywhere in this package).
"""
__all__ = ["api", "gp", "_lib"]
