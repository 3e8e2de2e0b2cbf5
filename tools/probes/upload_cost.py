#!/usr/bin/env python3
"""What does the upload of 3 x 80 MB from pageable host arrays cost in a fresh process -- first call, again with the same arrays,
again with fresh arrays?  (the Fortran API's first gadf_fit waits ~16 ms for it, bench.py's setup leg measures 5 ms in a process that
has uploaded the same arrays before)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from gadfit_amd import _lib
n = 10_000_000
def fresh():
    return np.random.default_rng(1).random(n), np.ones(n), np.full(n, 2.0)
pos = [0, n]
c = _lib.Context(0)
a = fresh()
for label, arrs in (('first upload in the process', a), ('same arrays again', a), ('fresh arrays', fresh()), ('fresh arrays', fresh())):
    t0 = time.perf_counter(); c.set_data(arrs[0], arrs[1], arrs[2], pos); dt = time.perf_counter() - t0
    print("%-30s %7.2f ms" % (label, 1e3 * dt), flush=True)
c2 = _lib.Context(0)
b = fresh()
t0 = time.perf_counter(); c2.set_data(b[0], b[1], b[2], pos); print('%-30s %7.2f ms' % ('second context, fresh arrays', 1e3 * (time.perf_counter() - t0)))
c.close(); c2.close()
