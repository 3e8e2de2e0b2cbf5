#!/usr/bin/env python3
"""A/B of whole LM iterations at BASELINE config 2 (4-exponential, 8 active, N = 1e7: the launch chain, not the in-kernel
tail, follows the fused kernel) under library switches.  usage: ab_cfg2_fit.py "NAME=VAL,..." ...  ("" = defaults); each variant
is measured three times, interleaved."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gadfit_amd import _lib
from gadfit_amd.ad import trace_model
from tests import models as M


def main():
    n = int(os.environ.get('AB_POINTS', '10000000'))
    x, y, s = M.make_single(M.exp4_numpy, M.EXP4_TRUTH, n, 0.0, 100.0)
    t = trace_model(M.model_exp4, 8)
    start = M.start_values(M.EXP4_TRUTH).reshape(1, 8); act = list(range(8))
    specs = sys.argv[1:] or ['']
    res = {sp: [] for sp in specs}
    for rnd in range(3):
        for sp in specs:
            for k in [k for k in os.environ if k.startswith('GADFIT_HIP_') and k != 'GADFIT_HIP_CACHE']:
                del os.environ[k]
            for kv in [p for p in sp.split(',') if p]:
                a, b = kv.split('='); os.environ['GADFIT_HIP_' + a] = b
            c = _lib.Context(0)
            c.set_model(t); c.set_data(x, y, 1.0 / s, [0, n])
            for _ in range(12):
                c.fit(start.copy(), act, [0] * 8, lambda_=1.0, max_iter=7)
            t0 = time.perf_counter(); it = 0
            for _ in range(30):
                _, r = c.fit(start.copy(), act, [0] * 8, lambda_=1.0, max_iter=7); it += r.iterations
            res[sp].append(round(1e3 * (time.perf_counter() - t0) / it, 4))
            c.close()
    for sp in specs:
        print(json.dumps({'variant': sp or 'default', 'ms_per_lm_iteration': res[sp]}), flush=True)


if __name__ == '__main__':
    main()
