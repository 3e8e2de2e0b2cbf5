#!/usr/bin/env python3
"""Does a fresh box speed up while it works?  The headline fits, 100 LM iterations per line, for ~12 s from the first GPU process of a
box: ms per iteration and the fused kernel's HIP-event average per block.  usage: warm_drift.py [seconds]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from gadfit_amd import _lib
from gadfit_amd.ad import trace_model
from tests import models as M

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 12.0
n = 10_000_000
truth = M.gauss8_truth()
x, y, s = M.make_single(M.gauss8_numpy, truth, n, 0.0, 100.0)
ctx = _lib.Context(0)
ctx.set_placement_after(0)
ctx.set_model(trace_model(M.model_gauss8, 32)); ctx.set_data(x, y, 1.0 / s, [0, n])
act = list(range(32)); glob = [0] * 32; start = M.start_values(truth).reshape(1, 32)
t_begin = time.perf_counter()
k = 0
while time.perf_counter() - t_begin < secs:
    ctx.reset_timers()
    t0 = time.perf_counter()
    for _ in range(10):
        ctx.fit(start, act, glob, lambda_=1.0, max_iter=10)
    dt = (time.perf_counter() - t0) / 100
    tm = ctx.timers()
    if k % 10 == 0 or k < 5:
        print('t = %5.2f s  ms per iteration %.4f  kernel %.4f' % (time.perf_counter() - t_begin, 1e3 * dt, 1e3 * tm[0] / max(1.0, tm[6])), flush=True)
    k += 1
print('placement', [round(v, 4) for v in ctx.placement()])
ctx.close()
