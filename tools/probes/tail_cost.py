#!/usr/bin/env python3
"""What the in-kernel tail of the fused kernel costs at the headline size: the kernel back to back without its tail (gfh_time_kernel:
workgroup partials only) against the same kernel inside gfh_fit (tail_mode 2: reduction over workgroups, assembly, mailbox), same
context, same buffers, alternating.  usage: tail_cost.py [N]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from gadfit_amd import _lib
from gadfit_amd.ad import trace_model
from tests import models as M

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
truth = M.gauss8_truth()
x, y, s = M.make_single(M.gauss8_numpy, truth, n, 0.0, 100.0)
ctx = _lib.Context(0)
ctx.set_placement_after(0)
ctx.set_model(trace_model(M.model_gauss8, 32)); ctx.set_data(x, y, 1.0 / s, [0, n])
act = list(range(32)); glob = [0] * 32; start = M.start_values(truth).reshape(1, 32)
jac, dim = ctx.jacobian_indices(act, glob)
ctx.sweep(start, act, jac, dim)
ctx.time_kernel(5, 80)
ctx.set_timer_detail(2)
for r in range(4):
    a = ctx.time_kernel(5, 100)
    ctx.reset_timers()
    t0 = time.perf_counter()
    for _ in range(10):
        ctx.fit(start, act, glob, lambda_=1.0, max_iter=10)
    wall = (time.perf_counter() - t0) / 100
    tm = ctx.timers()
    print(json.dumps({'round': r, 'kernel_no_tail_ms': round(a, 4), 'kernel_in_fit_ms': round(1e3 * tm[0] / max(1.0, tm[6]), 4), 'lm_iteration_wall_ms': round(1e3 * wall, 4),
                      'launches': int(tm[6])}), flush=True)
ctx.close()
