#!/usr/bin/env python3
"""BASELINE config 2 (4-exponential, 8 active parameters, N = 1e7) through gfh_fit: wall time per LM iteration; under
`rocprofv3 --kernel-trace` + tools/probes/timeline.py it shows which launches an iteration is made of."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from gadfit_amd import _lib
from gadfit_amd.ad import trace_model
from tests import models as M

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
x, y, s = M.make_single(M.exp4_numpy, M.EXP4_TRUTH, n, 0.0, 100.0)
ctx = _lib.Context(0)
ctx.set_model(trace_model(M.model_exp4, 8))
ctx.set_data(x, y, 1.0 / s, [0, n])
start = M.start_values(M.EXP4_TRUTH).reshape(1, 8)
act = list(range(8)); glob = [0] * 8
for _ in range(12):
    ctx.fit(start.copy(), act, glob, lambda_=1.0, max_iter=5)
t0 = time.perf_counter(); k = 0
for _ in range(40):
    _, r = ctx.fit(start.copy(), act, glob, lambda_=1.0, max_iter=5)
    k += r.iterations
dt = time.perf_counter() - t0
print('cfg2 N=%d: %.4f ms per LM iteration (%d iterations; last fit: %d sweeps, %d chi2 passes, %d look-ahead)' % (n, 1e3 * dt / k, k, r.n_sweeps, r.n_chi2, r.n_lookahead))
ctx.close()
