#!/usr/bin/env python3
"""What a small fit's whole cycle costs besides the fit: context creation, model, data, destruction (gadf_init ... gadf_close of a
batch of small fits, tests/fortran/bench_many_small_fits.F90)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from gadfit_amd import _lib
from gadfit_amd.ad import trace_model, exp
n = 1000
x = np.linspace(0.005, 9.995, n); y = 3.0 * np.exp(-((x - 4.5) / 0.8) ** 2) + 0.5; w = np.ones(n)
tape = trace_model(lambda p, x: p[0] * exp(-((x - p[1]) / p[2]) ** 2) + p[3], 4)
start = np.array([[2.5, 4.3, 1.0, 0.3]])
acc = {}
def lap(name, t0):
    t1 = time.perf_counter(); acc.setdefault(name, []).append(1e3 * (t1 - t0)); return t1
for k in range(12):
    t = time.perf_counter()
    c = _lib.Context(0); t = lap('create', t)
    c.set_keep_jacobian(2)
    c.set_model(tape); t = lap('set_model', t)
    c.set_data(x, y, w, [0, n]); t = lap('set_data', t)
    p, r = c.fit(start.copy(), [0, 1, 2, 3], [0] * 4, lambda_=1.0, max_iter=30); t = lap('fit', t)
    c.close(); t = lap('close', t)
for k, v in acc.items():
    print('%-10s first %8.3f ms   later (median) %8.3f ms' % (k, v[0], sorted(v[1:])[len(v[1:]) // 2]))
