#!/usr/bin/env python3
"""The part of an LM iteration that is not the N-sized kernel: gfh_fit at the headline model (32 active parameters) on 512 points --
every iteration is one launch of the fused kernel (one workgroup, one pass), its tail and host mailbox, the damped 32 x 32 solve
on the host and the next launch.  Wall time per iteration = the launch-to-launch latency floor of the look-ahead schedule."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from gadfit_amd import _lib
from gadfit_amd.ad import trace_model
from tests import models as M
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
truth = M.gauss8_truth()
x, y, s = M.make_single(M.gauss8_numpy, truth, n, 0.0, 100.0)
c = _lib.Context(0)
c.set_model(trace_model(M.model_gauss8, 32))
c.set_data(x, y, 1 / s, [0, n])
act = list(range(32)); glob = [0] * 32
start = M.start_values(truth).reshape(1, 32)
c.set_lookahead(True)
for _ in range(20):
    c.fit(start.copy(), act, glob, lambda_=1.0, max_iter=20)
for rep in range(3):
    t0 = time.perf_counter(); it = 0
    for _ in range(50):
        _, r = c.fit(start.copy(), act, glob, lambda_=1e3, lam_down=1.0, max_iter=20)     # lambda fixed and large: every step accepted
        it += r.iterations
    dt = time.perf_counter() - t0
    print('N = %d: %.2f us per LM iteration (%d iterations, %d sweeps, %d chi2 in the last fit)' % (n, 1e6 * dt / it, it, r.n_sweeps, r.n_chi2 - r.n_lookahead), flush=True)
c.close()
