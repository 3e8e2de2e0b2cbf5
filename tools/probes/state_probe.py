#!/usr/bin/env python3
"""Is the fused kernel slower inside the LM loop than launched back to back?  One process, one context, alternating:
  M1  100 launches back to back (gfh_time_kernel(5, 100))
  M2  100 single launches, the host synchronising after each (a ~20 us idle gap between launches)
  M3  the LM loop itself (gfh_fit, 100 iterations: HIP events around every launch, gfh_get_timers)
Optionally (argv[1] == 'torch') torch is imported and 2 GiB allocated first, as bench.py does."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == 'torch':
    import torch
    ballast = [torch.empty(1 << 27, dtype=torch.float64, device='cuda') for _ in range(2)]
import numpy as np
from gadfit_amd import _lib
from gadfit_amd.ad import trace_model
from tests import models as M

N = 10_000_000
truth = M.gauss8_truth()
x, y, s = M.make_single(M.gauss8_numpy, truth, N, 0.0, 100.0)
c = _lib.Context(0)
c.set_model(trace_model(M.model_gauss8, 32))
c.set_data(x, y, 1 / s, [0, N])
act = list(range(32)); glob = [0] * 32
jac, dim = c.jacobian_indices(act, glob)
start = M.start_values(truth).reshape(1, 32)
c.set_lookahead(True)
for _ in range(8):
    c.fit(start.copy(), act, glob, lambda_=1.0, max_iter=10)
c.sweep(start, act, jac, dim)
c.time_kernel(5, 100)
for rnd in range(5):
    m1 = c.time_kernel(5, 100)
    t = [c.time_kernel(5, 1) for _ in range(100)]
    m2 = float(np.mean(t[10:]))
    c.reset_timers()
    for _ in range(10):
        c.fit(start.copy(), act, glob, lambda_=1.0, max_iter=10)
    tm = c.timers()
    m3 = 1e3 * tm[0] / max(1.0, tm[6])
    c.sweep(start, act, jac, dim)
    print('round %d: back-to-back %.4f ms   single launches %.4f ms   LM loop %.4f ms (%d launches)' % (rnd, m1, m2, m3, int(tm[6])), flush=True)
c.close()
