#!/usr/bin/env python3
"""Can a trickle of small launches during an idle gap keep the part out of the clock ramp that slows the first ~40 launches after
it?  A second context (its own stream) runs a small chi2 (N = 65536) every `period` ms through the gap; then 20 fused sweeps of the
headline workload are timed.   python tools/probes/keepalive_probe.py"""
import os, sys, time, threading
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
tape = trace_model(M.model_gauss8, 32)
ctx = _lib.Context(0); ctx.set_model(tape); ctx.set_data(x, y, w, [0, n])
active = list(range(32)); start = M.start_values(truth).reshape(1, 32)
jac, dim = ctx.jacobian_indices(active, [0] * 32)
for _ in range(100):
    ctx.sweep(start, active, jac, dim)
ctx.set_timer_detail(2)

def make_small(m):
    c = _lib.Context(0); c.set_model(tape); c.set_data(x[:m], y[:m], w[:m], [0, m]); c.chi2(start); return c
small = {m: make_small(m) for m in (65536, 1048576)}

def sweeps(label):
    d = []
    t0 = time.perf_counter()
    for i in range(20):
        ctx.sweep(start, active, jac, dim)
        d.append(ctx.timer_spread()[2] * 1e3)
    wall = (time.perf_counter() - t0) * 1e3
    d = np.array(d)
    print('%-52s 20 sweeps %.2f ms wall | kernel ms: [0,5) %.3f  [5,10) %.3f  [10,20) %.3f' % (label, wall, d[:5].mean(), d[5:10].mean(), d[10:].mean()), flush=True)

def gap(seconds, period_ms, m, burst=1):
    stop = threading.Event(); cnt = [0]
    def keeper():
        c = small[m]
        while not stop.is_set():
            for _ in range(burst):
                c.chi2(start)
            cnt[0] += burst
            if period_ms > 0:
                time.sleep(period_ms * 1e-3)
    th = None
    if period_ms >= 0:
        th = threading.Thread(target=keeper); th.start()
    time.sleep(seconds)
    if th:
        stop.set(); th.join()
    return cnt[0]

for rnd in range(2):
    gap(0.5, -1, 65536); sweeps('idle 0.5 s')
    for m in (65536, 1048576):
        for period in (10.0, 2.0, 0.5, 0.0):
            k = gap(0.5, period, m); sweeps('0.5 s gap, chi2(N=%d) every %.1f ms (%d launches)' % (m, period, k))
    k = gap(0.5, 5.0, 1048576, burst=8); sweeps('0.5 s gap, 8 x chi2(N=1048576) every 5 ms (%d)' % k)
    ctx.sweep(start, active, jac, dim)
    for _ in range(60):
        ctx.sweep(start, active, jac, dim)
    sweeps('steady state')
ctx.close()
