#!/usr/bin/env python3
"""BASELINE config 3 (global fit: 64 datasets x 1e5 points, 4 local + 3 global parameters, dim 259) through gfh_fit:
wall time per LM iteration.  Under `rocprofv3 --kernel-trace --stats` it shows which kernels an iteration is made of."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from gadfit_amd import _lib
from gadfit_amd.ad import trace_model
from tests import models as M


def main():
    nd = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    xs, ys, ss, truths = M.make_global7(nd, 100_000)
    pars = np.array([M.start_values(t) for t in truths]); pars[:, 4:] = M.start_values(M.GLOBAL7_TAUS)
    ctx = _lib.Context(0)
    ctx.set_model(trace_model(M.model_global7, 7))
    pos = np.concatenate([[0], np.cumsum([a.size for a in xs])])
    ctx.set_data(np.concatenate(xs), np.concatenate(ys), np.concatenate([1 / s for s in ss]), pos)
    act = list(range(7)); glob = [0, 0, 0, 0, 1, 1, 1]
    for _ in range(12):
        ctx.fit(pars.copy(), act, glob, lambda_=1.0, max_iter=5)
    t0 = time.perf_counter(); n = 0
    for _ in range(20):
        _, r = ctx.fit(pars.copy(), act, glob, lambda_=1.0, max_iter=5)
        n += r.iterations
    dt = time.perf_counter() - t0
    print('cfg3 nd=%d dim=%d: %.4f ms per LM iteration (%d iterations, %d sweeps + %d chi2 passes in the last fit)'
          % (nd, r.dim, 1e3 * dt / n, n, r.n_sweeps, r.n_chi2 - r.n_lookahead))
    ctx.close()


if __name__ == '__main__':
    main()
