#!/usr/bin/env python3
"""Does a small upload made earlier (at context creation, say) take the one-time 11 ms off the first large upload of a process?
usage: upload_warm.py <n_small points, 0 = none>"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from gadfit_amd import _lib
n = 10_000_000
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 0
c = _lib.Context(0)
if ns:
    s = np.ones(ns)
    t0 = time.perf_counter(); c.set_data(s, s, s, [0, ns]); print("small upload of 3 x %d doubles %7.2f ms" % (ns, 1e3 * (time.perf_counter() - t0)), flush=True)
x, y, w = np.random.default_rng(1).random(n), np.ones(n), np.full(n, 2.0)
for label in ('first large upload', 'again'):
    t0 = time.perf_counter(); c.set_data(x, y, w, [0, n]); print("%-30s %7.2f ms" % (label, 1e3 * (time.perf_counter() - t0)), flush=True)
c.close()
