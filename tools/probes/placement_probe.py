#!/usr/bin/env python3
"""The Jacobian buffer's placement (gfh_set_placement_tries): a row of fresh contexts at the headline size, alternately taking the
first allocation and the best of N, some contexts kept alive so that later ones get other pages.  Prints the fused kernel's and
the plain sweep's time per context and what the placement saw."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from gadfit_amd import _lib
from gadfit_amd.ad import trace_model
from tests import models as M
N = 10_000_000
tries = int(sys.argv[1]) if len(sys.argv) > 1 else 6
truth = M.gauss8_truth()
x, y, s = M.make_single(M.gauss8_numpy, truth, N, 0.0, 100.0)
tape = trace_model(M.model_gauss8, 32)
pars = np.array([M.start_values(truth)]); active = list(range(32))
keep = []
res = {1: [], tries: []}
for rnd in range(16):
    n = 1 if rnd % 2 == 0 else tries
    c = _lib.Context(0)
    c.set_placement_tries(n); c.set_placement_after(0)
    c.set_model(tape); c.set_data(x, y, 1 / s, [0, N])
    jac, dim = c.jacobian_indices(active, [0] * 32)
    for _ in range(40):
        c.sweep(pars, active, jac, dim)
    c.time_kernel(4, 60); ts = c.time_kernel(4, 100); t = c.time_kernel(5, 100)
    res[n].append(t * 1e3)
    print('ctx %2d tries %d: plain %.1f us  fused %.1f us | placement (ms, kept first): %s' % (rnd, n, ts * 1e3, t * 1e3, ['%.3f' % v for v in c.placement()]), flush=True)
    if rnd % 4 == 3:
        keep.append(c)
    else:
        c.close()
for n, v in res.items():
    print('tries %d: fused mean %.1f us, min %.1f, max %.1f' % (n, np.mean(v), min(v), max(v)))
