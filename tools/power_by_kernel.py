#!/usr/bin/env python3
"""Socket power and shader clock (rocm-smi) while ONE kernel of the headline workload runs back to back for ~2 s each: plain sweep,
fused sweep+Gram (Jacobian stored), fused without the store, chi2, omega+J^T omega.  Which part of the pass meets the power limit?"""
import json
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gadfit_amd import _lib
from gadfit_amd.ad import trace_model
from tests import models as M


def sample(stop, out):
    while not stop.is_set():
        try:
            d = json.loads(subprocess.run(['rocm-smi', '--showclocks', '--showpower', '--json'], capture_output=True, text=True, timeout=5).stdout)
            c = d.get('card0', d)
            out.append((float(c['Current Socket Graphics Package Power (W)']), int(c['sclk clock speed:'].strip('()Mhz'))))
        except Exception:
            pass
        time.sleep(0.1)


def main():
    n = 10_000_000
    truth = M.gauss8_truth()
    x, y, s = M.make_single(M.gauss8_numpy, truth, n, 0.0, 100.0)
    tape = trace_model(M.model_gauss8, 32)
    pars = M.start_values(truth).reshape(1, 32); act = list(range(32))
    for label, which, keep in [('plain sweep (gfh_k_sweep)', 4, 1), ('fused sweep+Gram, J stored', 5, 1), ('fused sweep+Gram, no J store', 5, 0),
                               ('chi2', 2, 1), ('omega + J^T omega', 6, 1)]:
        os.environ['GADFIT_HIP_KEEP_J'] = str(keep)
        c = _lib.Context(0)
        c.set_model(tape); c.set_data(x, y, 1 / s, [0, n])
        jac, dim = c.jacobian_indices(act, [0] * 32)
        JTJ, JTr, chi2 = c.sweep(pars, act, jac, dim)
        c.omega(pars, np.linspace(0.1, 0.4, dim))
        ms0 = c.time_kernel(which, 50)
        reps = max(50, int(2.5e3 / ms0))
        stop = threading.Event(); got = []
        th = threading.Thread(target=sample, args=(stop, got)); th.start()
        ms = c.time_kernel(which, reps)
        stop.set(); th.join()
        body = got[len(got) // 3:] or got
        print('%-30s %.4f ms  power %4.0f W  sclk %4.0f MHz  (%d samples)' % (label, ms, np.mean([g[0] for g in body]), np.mean([g[1] for g in body]), len(body)), flush=True)
        c.close()
        time.sleep(1.0)


if __name__ == '__main__':
    main()
