#!/usr/bin/env python3
"""One-time costs around a fit at the headline size (N = 1e7 x 32 parameters): data hand-over (gfh_set_data), weights, kernel
load from the cache, first pass.  A ten-iteration fit is 5-6 ms, so these decide what a single fit costs end to end."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from gadfit_amd import _lib
from gadfit_amd.ad import trace_model
from tests import models as M


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    truth = M.gauss8_truth()
    x, y, s = M.make_single(M.gauss8_numpy, truth, n, 0.0, 100.0)
    tape = trace_model(M.model_gauss8, 32)
    act = list(range(32)); start = M.start_values(truth).reshape(1, 32)
    t = [time.perf_counter()]
    c = _lib.Context(0); t.append(time.perf_counter())
    c.set_model(tape); t.append(time.perf_counter())
    c.set_data(x, y, s, [0, n]); t.append(time.perf_counter())
    c.set_data(x, y, s, [0, n]); t.append(time.perf_counter())
    c.init_weights(4); t.append(time.perf_counter())
    jac, dim = c.jacobian_indices(act, [0] * 32)
    c.sweep(start, act, jac, dim); t.append(time.perf_counter())
    c.sweep(start, act, jac, dim); t.append(time.perf_counter())
    names = ['gfh_create', 'gfh_set_model', 'gfh_set_data (first)', 'gfh_set_data (again)', 'gfh_init_weights', 'first gfh_sweep (kernel load)', 'second gfh_sweep']
    for nm, a, b in zip(names, t, t[1:]):
        print('%-32s %9.3f ms' % (nm, 1e3 * (b - a)))
    c.close()


if __name__ == '__main__':
    main()
