#!/usr/bin/env python3
"""One CPU "image" of bench.py's all-cores cpu_baseline leg (TEST / MEASUREMENT INFRASTRUCTURE, never a
product path): the reference parallelises over coarray images, each sweeping its contiguous share of the
points (gadfit.F90:977-1002); this worker is one such image running the oracle on its share.

  python oracle/bench_worker.py n_total begin count iterations start_epoch

Waits until `start_epoch` (so all images start together), runs `iterations` x (sweep + chi2) on the slice
and prints one JSON line {"t0": ..., "t1": ...} with wall-clock times of its compute phase."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    n_total, begin, count, iters = (int(a) for a in sys.argv[1:5])
    start_epoch = float(sys.argv[5])
    import numpy as np
    from gadfit_amd.ad import trace_model
    from oracle import binding as orc
    from tests import models as M
    truth = M.gauss8_truth()
    x, y, s = M.make_single_slice(M.gauss8_numpy, truth, n_total, begin, count, 0.0, 100.0)
    tape = trace_model(M.model_gauss8, 32)
    p = orc.OracleProblem(tape, [x], [y], [1.0 / s], [M.start_values(truth)], list(range(32)), [0] * 32)
    while time.time() < start_epoch:
        time.sleep(0.001)
    t0 = time.time()
    for _ in range(iters):
        p.sweep(); p.chi2()
    t1 = time.time()
    print(json.dumps({'t0': t0, 't1': t1, 'count': count}))


if __name__ == '__main__':
    main()
