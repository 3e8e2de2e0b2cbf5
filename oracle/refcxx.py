"""ctypes binding of oracle/_ref/libgadfit_refcxx.so: the REFERENCE'S OWN C++ automatic differentiation (compiled from where it
lies, oracle/Makefile) under this repository's caller oracle/ref_cxx_driver.cpp.  TEST INFRASTRUCTURE: only tests/ and bench.py's
cpu_baseline leg may import this."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, '_ref', 'libgadfit_refcxx.so')
GAUSS8, EXP4, INTEGRAL_SINGLE, GLOBAL7 = 0, 1, 2, 3
_LIB = None


def available():
    return os.path.exists(SO)


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(SO)
        _LIB.refcxx_chi2.restype = C.c_double
    return _LIB


def _dp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def sweep(model, x, y, sigma, pars, threads=1, want_J=True):
    """STEP 1 + 2 through gadfit::AdVar / returnSweep / dsyrk / dgemv.  Returns JTJ, JTres, res, J[n][p], (jacobian s, linalg s)."""
    x = np.ascontiguousarray(x, dtype=np.float64); y = np.ascontiguousarray(y, dtype=np.float64)
    s = np.ascontiguousarray(sigma, dtype=np.float64); p = np.ascontiguousarray(pars, dtype=np.float64).ravel()
    n = x.size; npar = lib().refcxx_n_pars(model)
    assert p.size == npar and y.size == n and s.size == n
    J = np.empty((n, npar)) if want_J else None
    res = np.empty(n) if want_J else None
    JTJ = np.empty((npar, npar)); JTr = np.empty(npar); sec = np.zeros(2)
    lib().refcxx_sweep(model, C.c_long(n), _dp(x), _dp(y), _dp(s), _dp(p), int(threads), _dp(J), _dp(res), _dp(JTJ), _dp(JTr), _dp(sec))
    return JTJ, JTr, res, J, (float(sec[0]), float(sec[1]))


def chi2(model, x, y, sigma, pars, threads=1):
    x = np.ascontiguousarray(x, dtype=np.float64); y = np.ascontiguousarray(y, dtype=np.float64)
    s = np.ascontiguousarray(sigma, dtype=np.float64); p = np.ascontiguousarray(pars, dtype=np.float64).ravel()
    sec = np.zeros(1)
    v = lib().refcxx_chi2(model, C.c_long(x.size), _dp(x), _dp(y), _dp(s), _dp(p), int(threads), _dp(sec))
    return float(v), float(sec[0])


def omega(model, x, sigma, pars, delta1, threads=1):
    """f''_delta1(x_i) / sigma_i through gadfit::AdVar's forward mode (lm_solver.cpp:360-380)"""
    x = np.ascontiguousarray(x, dtype=np.float64); s = np.ascontiguousarray(sigma, dtype=np.float64)
    p = np.ascontiguousarray(pars, dtype=np.float64).ravel(); d = np.ascontiguousarray(delta1, dtype=np.float64).ravel()
    out = np.empty(x.size)
    lib().refcxx_omega(model, C.c_long(x.size), _dp(x), _dp(s), _dp(p), _dp(d), int(threads), _dp(out))
    return out
