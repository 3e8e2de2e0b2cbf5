#!/usr/bin/env python3
"""What ends the power-management transient (first ~40 launches after an idle gap 20-35 % slower, tools/probes/transient.py)?
After 2 s idle: (a) nothing; (b) 20 ms of chi2 launches (FP64 work, no stores); (c) 20 ms of plain-sweep launches (the store
stream); (d) a fresh upload of the 240 MB of points (DMA only); then 60 fused sweeps, per-launch kernel time by index range."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from gadfit_amd import _lib
from gadfit_amd.ad import trace_model
from tests import models as M

n = 10_000_000
truth = M.gauss8_truth()
x, y, s = M.make_single(M.gauss8_numpy, truth, n, 0.0, 100.0)
w = 1 / s
ctx = _lib.Context(0)
ctx.set_model(trace_model(M.model_gauss8, 32))
ctx.set_data(x, y, w, [0, n])
active = list(range(32)); start = M.start_values(truth).reshape(1, 32)
jac, dim = ctx.jacobian_indices(active, [0] * 32)
for _ in range(100):
    ctx.sweep(start, active, jac, dim)
ctx.omega(start, np.zeros(dim))
ctx.set_timer_detail(2)          # every launch timed (the default samples one in eight)

def sweeps(label):
    d = []
    t0 = time.perf_counter()
    for i in range(60):
        ctx.sweep(start, active, jac, dim)
        d.append(ctx.timer_spread()[2] * 1e3)
    wall = (time.perf_counter() - t0) * 1e3
    d = np.array(d)
    print('%-34s first 60 sweeps %.2f ms wall | kernel ms: [0,5) %.3f  [5,10) %.3f  [10,20) %.3f  [20,40) %.3f  [40,60) %.3f'
          % (label, wall, d[:5].mean(), d[5:10].mean(), d[10:20].mean(), d[20:40].mean(), d[40:].mean()), flush=True)

for rnd in range(2):
    time.sleep(2.0); sweeps('(a) idle 2 s')
    time.sleep(2.0); t0 = time.perf_counter(); ctx.time_kernel(2, 200); print('    chi2 x200: %.1f ms' % ((time.perf_counter() - t0) * 1e3)); sweeps('(b) idle 2 s + 20 ms chi2')
    time.sleep(2.0); t0 = time.perf_counter(); ctx.time_kernel(4, 40); print('    plain sweep x40: %.1f ms' % ((time.perf_counter() - t0) * 1e3)); sweeps('(c) idle 2 s + 20 ms plain sweeps')
    time.sleep(2.0); t0 = time.perf_counter(); ctx.set_data(x, y, w, [0, n]); print('    set_data: %.1f ms' % ((time.perf_counter() - t0) * 1e3)); sweeps('(d) idle 2 s + upload')
ctx.close()
