"""CPU-reference calibration on the headline workload ("What's missing" 3).  Times, on identical seeded inputs,

  * the REFERENCE: oracle/_ref/libgadfit_refcxx.so -- the reference's own C++ AD and vendored dsyrk/dgemv compiled from where they
    lie (oracle/Makefile) under oracle/ref_cxx_driver.cpp, whose loop is LMsolver::computeLeftHandSide / computeRightHandSide /
    chi2 (c++/gadfit/lm_solver.cpp:286-346, 513-529), at 1 and 8 threads (OpenMP, as the reference parallelises);
  * the PORT: oracle/gadfit_oracle.c (the restatement of the Fortran side), 1 thread,

for (a) gauss8 with 32 active parameters (the headline model) and (b) the 4-exponential 8-parameter model (BASELINE config 2) at
N = 1e6 points, `iters` passes of (STEP 1 + STEP 2 + one chi2) each -- what one accepted LM iteration of the reference costs.
Writes profiles/r06_cpu_calibration.json; bench.py reads the ratio from there.  LMsolver::fit itself cannot be built here
(lm_solver.cpp needs spdlog, absent from the image; stand-in headers are against the rules): the lambda loop and the p x p
Cholesky it would add are O(p^3) per iteration, 1e-5 of the N-sized work timed here.

Usage: python tools/calibrate_cpu_baseline.py [N] [iters]"""
import json
import os
import platform
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from gadfit_amd.ad import trace_model      # noqa: E402
from oracle import binding as orc          # noqa: E402
from oracle import refcxx                  # noqa: E402
from tests import models as M              # noqa: E402


def one(name, mid, model, fn, truth, npar, n, iters):
    x, y, s = M.make_single(fn, truth, n, 0.0, 100.0)
    start = M.start_values(truth)
    out = {'model': name, 'active_params': npar, 'points': n, 'iterations': iters}
    for th in (1, 8):
        refcxx.sweep(mid, x[:20000], y[:20000], s[:20000], start, threads=th, want_J=False)
        t0 = time.perf_counter(); tj = tl = tc = 0.0
        for _ in range(iters):
            r = refcxx.sweep(mid, x, y, s, start, threads=th, want_J=False); tj += r[4][0]; tl += r[4][1]
            tc += refcxx.chi2(mid, x, y, s, start, threads=th)[1]
        dt = time.perf_counter() - t0
        out['reference_ns_%dt' % th] = 1e9 * dt / (n * iters)
        out['reference_split_ns_%dt' % th] = {'jacobian_loop': 1e9 * tj / (n * iters), 'dsyrk_dgemv': 1e9 * tl / (n * iters), 'chi2': 1e9 * tc / (n * iters)}
    tape = trace_model(model, npar)
    p = orc.OracleProblem(tape, [x], [y], [1.0 / s], [start], list(range(npar)), [0] * npar)
    t0 = time.perf_counter()
    for _ in range(iters):
        p.sweep(); p.chi2()
    out['port_ns_1t'] = 1e9 * (time.perf_counter() - t0) / (n * iters)
    out['ratio'] = out['port_ns_1t'] / out['reference_ns_1t']
    return out


def quadrature(n, iters):
    """BASELINE config 4's model (adaptive GK15 through AD) through the reference's own C++ integrate() and through the oracle"""
    from tests.golden import goldens as G
    a, b = 7.5, 0.8
    x = 0.05 + (10.0 - 0.05) * (np.arange(n) + 0.5) / n
    y = np.zeros(n); s = np.ones(n); start = np.array([a * 1.05, b * 0.95])
    import ctypes
    refcxx.lib().refcxx_set_rel_error.argtypes = [ctypes.c_double]
    refcxx.lib().refcxx_set_rel_error(1e-10)
    out = {'model': 'integral_single (pi int_0^x t^a exp(-b t^2) dt, GK15, rel 1e-10)', 'active_params': 2, 'points': n, 'iterations': iters}
    for th in (1, 8):
        t0 = time.perf_counter()
        for _ in range(iters):
            refcxx.sweep(refcxx.INTEGRAL_SINGLE, x, y, s, start, threads=th, want_J=False)
            refcxx.chi2(refcxx.INTEGRAL_SINGLE, x, y, s, start, threads=th)
        out['reference_ns_%dt' % th] = 1e9 * (time.perf_counter() - t0) / (n * iters)
    t = trace_model(G.model_integral_single, 2); t.set_integration(rel_error=1e-10)
    p = orc.OracleProblem(t, [x], [y], [1.0 / s], [start], [0, 1], [0, 0])
    t0 = time.perf_counter()
    for _ in range(iters):
        p.sweep(); p.chi2()
    out['port_ns_1t'] = 1e9 * (time.perf_counter() - t0) / (n * iters)
    out['ratio'] = out['port_ns_1t'] / out['reference_ns_1t']
    return out


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    rec = {'what': 'ns per point and LM iteration (STEP 1 + STEP 2 + one chi2): the reference\'s own C++ AD + vendored dsyrk/dgemv '
                   '(oracle/_ref/libgadfit_refcxx.so) against oracle/gadfit_oracle.c on identical inputs; ratio = port / reference at 1 thread',
           'host': {'machine': platform.machine(), 'cpus': os.cpu_count(), 'cpu_model': next((l.split(':', 1)[1].strip() for l in open('/proc/cpuinfo') if l.startswith('model name')), None)},
           'reference_sources': 'c++/gadfit/{automatic_differentiation,fit_function,exceptions,lapack_fallback,numerical_integration}.cpp, g++ -std=c++20 -O2 -fopenmp; loop = lm_solver.cpp:286-346, 513-529',
           'not_timed': 'LMsolver::fit\'s lambda loop and Cholesky (lm_solver.cpp needs spdlog: unbuildable here)',
           'models': [one('gauss8', refcxx.GAUSS8, M.model_gauss8, M.gauss8_numpy, M.gauss8_truth(), 32, n, iters),
                      one('exp4', refcxx.EXP4, M.model_exp4, M.exp4_numpy, M.EXP4_TRUTH, 8, n, iters),
                      quadrature(max(2000, n // 50), iters)]}
    path = os.path.join(ROOT, 'profiles', 'r06_cpu_calibration.json')
    json.dump(rec, open(path, 'w'), indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == '__main__':
    main()
