"""ctypes wrapper for tests/host_logic/libhostlogic.so (product search logic on the CPU oracle)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host_logic")
_lib = None


def lib():
    global _lib
    if _lib is None:
        subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)
        _lib = ctypes.CDLL(os.path.join(_HERE, "libhostlogic.so"))
        dp = ctypes.POINTER(ctypes.c_double)
        _lib.hostlogic_calibrate.restype = ctypes.c_int
        _lib.hostlogic_calibrate.argtypes = [
            dp, dp, ctypes.c_double, ctypes.c_size_t, ctypes.c_longlong, ctypes.c_double, ctypes.c_int,
            ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_uint8), dp, dp,
            ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(ctypes.c_longlong)]
        _lib.hostlogic_solve_fixed.restype = ctypes.c_int
        _lib.hostlogic_solve_fixed.argtypes = [
            dp, dp, ctypes.c_double, ctypes.c_size_t, ctypes.c_double,
            ctypes.POINTER(ctypes.c_uint8), dp, ctypes.POINTER(ctypes.c_longlong),
            ctypes.POINTER(ctypes.c_longlong)]
    return _lib


def _dp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double)) if a is not None else None


def calibrate(scores, gamma_or_costs, target, max_iter=60, spec_depth=2, force_exact=False):
    s = np.ascontiguousarray(scores, dtype=np.float64)
    n = s.shape[0]
    if np.isscalar(gamma_or_costs):
        costs, gamma = None, float(gamma_or_costs)
        total = float(np.sum(np.full(max(n - 1, 0), gamma)))
    else:
        costs, gamma = np.ascontiguousarray(gamma_or_costs, dtype=np.float64), 0.0
        total = float(np.sum(costs))
    sol = np.zeros(n, dtype=np.uint8)
    pen, val = ctypes.c_double(), ctypes.c_double()
    cnt = ctypes.c_longlong()
    info = (ctypes.c_longlong * 16)()
    rc = lib().hostlogic_calibrate(_dp(s), _dp(costs), gamma, n, int(target), total, int(max_iter),
                                   int(spec_depth), int(force_exact),
                                   sol.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), ctypes.byref(pen),
                                   ctypes.byref(val), ctypes.byref(cnt), info)
    assert rc == 0, rc
    keys = ["path", "evaluations", "passes", "zone_iters", "n_diff", "probe_calls", "window_calls",
            "exact_calls", "exact_lambdas", "maps", "spine_calls", "compact_n", "pilot_calls"]
    return pen.value, sol, val.value, cnt.value, dict(zip(keys, [int(x) for x in info]))


def solve_fixed(scores, gamma_or_costs, lam):
    s = np.ascontiguousarray(scores, dtype=np.float64)
    n = s.shape[0]
    if np.isscalar(gamma_or_costs):
        costs, gamma = None, float(gamma_or_costs)
    else:
        costs, gamma = np.ascontiguousarray(gamma_or_costs, dtype=np.float64), 0.0
    sol = np.zeros(n, dtype=np.uint8)
    val = ctypes.c_double()
    cnt = ctypes.c_longlong()
    info = (ctypes.c_longlong * 16)()
    rc = lib().hostlogic_solve_fixed(_dp(s), _dp(costs), gamma, n, float(lam),
                                     sol.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), ctypes.byref(val),
                                     ctypes.byref(cnt), info)
    assert rc == 0, rc
    return sol, val.value, cnt.value, {"path": int(info[0]), "n_diff": int(info[4])}
