#!/usr/bin/env python3
"""Does the fused kernel's speed depend on where its buffers land?  One process, the same workload (N points x 32 active) in a
row of fresh contexts; optionally a ballast allocation of BALLAST_MB megabytes is made (and kept) before every context, which shifts
the placement of everything after it.  Prints the fused kernel's time per context."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
from gadfit_amd import _lib
from gadfit_amd.ad import trace_model
from tests import models as M


def main():
    n = int(os.environ.get('AB_POINTS', '10000000'))
    rounds = int(os.environ.get('ROUNDS', '8'))
    ballast_mb = [int(v) for v in os.environ.get('BALLAST_MB', '0').split(',')]
    truth = M.gauss8_truth()
    x, y, s = M.make_single(M.gauss8_numpy, truth, n, 0.0, 100.0)
    tape = trace_model(M.model_gauss8, 32)
    pars = M.start_values(truth).reshape(1, 32); act = list(range(32))
    hip = C.CDLL('libamdhip64.so')
    keep = []
    for r in range(rounds):
        mb = ballast_mb[r % len(ballast_mb)]
        if mb:
            p = C.c_void_p()
            assert hip.hipMalloc(C.byref(p), C.c_size_t(mb << 20)) == 0
            keep.append(p)
        c = _lib.Context(0)
        c.set_model(tape); c.set_data(x, y, 1 / s, [0, n])
        jac, dim = c.jacobian_indices(act, [0] * 32)
        c.sweep(pars, act, jac, dim)
        ms = min(c.time_kernel(5, 30) for _ in range(3))
        print('context %d (ballast %d MB before it): fused %.4f ms = %.0f GB/s' % (r, mb, ms, 288 * n / (ms * 1e-3) / 1e9), flush=True)
        c.close()


if __name__ == '__main__':
    main()
