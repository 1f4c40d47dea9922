"""Convenience wrappers around oracle/oracle_driver (the plain-C restatement).  TEST INFRASTRUCTURE ONLY."""
import numpy as np

from . import refrun


def _need():
    if not refrun.have_port():
        raise ImportError("oracle/oracle_driver is not built (make -C oracle port)")


def kern(terms, X, X2, covGrad, covGrad2):
    _need()
    arrays = dict(refrun.kern_arrays(terms))
    arrays.update({"X": X, "X2": X2, "covGrad": covGrad, "covGrad2": covGrad2})
    return refrun.run_port("kern", arrays)


def gp(terms, X, y, Xstar=None, scale=None, bias=None, exact_trans=False, dump=False):
    _need()
    arrays = dict(refrun.kern_arrays(terms))
    arrays.update({"X": X, "y": y, "exact_trans": 1.0 if exact_trans else 0.0, "dump_matrices": 1.0 if dump else 0.0})
    if Xstar is not None:
        arrays["Xstar"] = Xstar
    if scale is not None:
        arrays["scale"] = np.asarray(scale).reshape(1, -1)
    if bias is not None:
        arrays["bias"] = np.asarray(bias).reshape(1, -1)
    return refrun.run_port("gp", arrays)


def time_update_k(terms, X, reps=1):
    _need()
    arrays = dict(refrun.kern_arrays(terms))
    arrays.update({"X": X, "reps": float(reps)})
    return refrun.run_port("time", arrays)


def chol(C, upper=True):
    _need()
    return refrun.run_port("chol", {"C": C, "upper": 1.0 if upper else 0.0})


def trsm(A, B, side, uplo, trans, diag, alpha):
    _need()
    flags = [side.upper() == "L", uplo.upper() == "U", trans.upper() != "N", diag.upper() == "U"]
    return refrun.run_port("trsm", {"A": A, "B": B, "flags": np.array(flags, dtype=float), "alpha": alpha})["X"]


def gplvm(terms, Y, X, regularise=True):
    """CGplvm objective + gradient (plain model) at latent points X: dict with ll, g, logdet, info, m."""
    _need()
    arrays = dict(refrun.kern_arrays(terms))
    arrays.update({"Y": Y, "X": X, "regularise": 1.0 if regularise else 0.0})
    return refrun.run_port("gplvm", arrays)
