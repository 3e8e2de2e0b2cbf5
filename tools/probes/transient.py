#!/usr/bin/env python3
"""Per-launch duration of the fused sweep+Gram kernel over a long run of back-to-back LM iterations
(cfg 5: 8 skewed Gaussians, 32 active, N=1e7): how the part's power management shapes the first launches
after an idle gap and where the steady state lies.  Prints launch index ranges with their mean duration."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from gadfit_amd import _lib
from gadfit_amd.ad import trace_model
from tests import models as M


def main():
    n = 10_000_000
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 600
    truth = M.gauss8_truth()
    x, y, s = M.make_single(M.gauss8_numpy, truth, n, 0.0, 100.0)
    ctx = _lib.Context(0)
    ctx.set_model(trace_model(M.model_gauss8, 32))
    ctx.set_data(x, y, 1 / s, [0, n])
    active = list(range(32)); start = M.start_values(truth).reshape(1, 32)
    jac, dim = ctx.jacobian_indices(active, [0] * 32)
    ctx.sweep(start, active, jac, dim)
    ctx.set_timer_detail(2)          # every launch timed (the default samples one in eight)
    for idle in (0.0, 2.0):
        time.sleep(idle)
        durs = []
        t0 = time.perf_counter()
        for i in range(iters):
            ctx.sweep(start, active, jac, dim)
            durs.append(ctx.timer_spread()[2] * 1e3)
        wall = (time.perf_counter() - t0) / iters * 1e3
        d = np.array(durs)
        edges = [0, 2, 5, 10, 20, 40, 80, 160, 320, 640, 1280, 2560]
        print('after %.0f s idle: wall %.4f ms per sweep call; kernel ms by launch index:' % (idle, wall))
        for a, b in zip(edges[:-1], edges[1:]):
            if a < len(d):
                print('  [%4d, %4d): mean %.4f  min %.4f  max %.4f' % (a, min(b, len(d)), d[a:b].mean(), d[a:b].min(), d[a:b].max()))
    ctx.close()


if __name__ == '__main__':
    main()
