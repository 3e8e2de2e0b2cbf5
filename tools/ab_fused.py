#!/usr/bin/env python3
"""A/B timing of the fused sweep+Gram kernel (cfg 5: 8 skewed Gaussians, 32 active, N=1e7) under
different generator/layout switches.  Each variant is a fresh context (the switches are read from the
environment at gfh_create); HIP-event time per launch of the fused kernel and of the plain sweep.
usage: ab_fused.py "NAME=VAL,NAME=VAL" "NAME=VAL" ...   (names without the GADFIT_HIP_ prefix; "" = defaults)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gadfit_amd import _lib
from gadfit_amd.ad import trace_model
from tests import models as M


def main():
    n = int(os.environ.get('AB_POINTS', '10000000'))
    truth = M.gauss8_truth()
    x, y, s = M.make_single(M.gauss8_numpy, truth, n, 0.0, 100.0)
    tape = trace_model(M.model_gauss8, 32)
    pars = M.start_values(truth).reshape(1, 32)
    active = list(range(32))
    ref = None
    for spec in sys.argv[1:] or ['']:
        kv = dict(p.split('=') for p in spec.split(',') if p)
        for k in [k for k in os.environ if k.startswith('GADFIT_HIP_') and k not in ('GADFIT_HIP_CACHE',)]:
            del os.environ[k]
        for k, v in kv.items():
            os.environ['GADFIT_HIP_' + k] = v
        ctx = _lib.Context(0)
        ctx.set_model(tape)
        ctx.set_data(x, y, 1 / s, [0, n])
        jac, dim = ctx.jacobian_indices(active, [0] * 32)
        JTJ, JTr, chi2 = ctx.sweep(pars, active, jac, dim)
        if ref is None:
            ref = (JTJ, chi2)
        err = float(np.max(np.abs(JTJ - ref[0]) / np.sqrt(np.outer(np.diag(ref[0]), np.diag(ref[0])))))
        out = {'variant': spec or 'default', 'chi2_rel_diff': abs(chi2 - ref[1]) / ref[1], 'JTJ_rel_diff': err}
        for label, which in [('fused', 5), ('sweep_only', 4)]:
            reps = int(os.environ.get('AB_REPS', '10'))
            ms = min(ctx.time_kernel(which, reps) for _ in range(3))
            out[label + '_ms'] = round(ms, 4)
            out[label + '_GBps'] = round(288 * n / (ms * 1e-3) / 1e9, 1)
        ctx.close()
        print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
