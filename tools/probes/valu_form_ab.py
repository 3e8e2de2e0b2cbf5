#!/usr/bin/env python3
"""A/B of the fused kernel's VALU form (BASELINE configs 2 and 3) inside ONE process: the variants' contexts are created in turn, several
times over (every context draws fresh pages for its Jacobian: the placement spread is part of what is averaged), medians per variant.
usage: valu_form_ab.py [cfg] [rounds]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from gadfit_amd import _lib
from gadfit_amd.ad import trace_model
from tests import models as M

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
VARIANTS = [dict(GADFIT_HIP_VALU_AHEAD=a, GADFIT_HIP_GB_TARGET=t) for a in ('1', '2') for t in ('256', '512')]
if cfg == 2:
    x, y, s = M.make_single(M.exp4_numpy, M.EXP4_TRUTH, 10_000_000, 0.0, 100.0)
    tape = trace_model(M.model_exp4, 8); xs, ys, ws = [x], [y], [1 / s]
    pars = M.start_values(M.EXP4_TRUTH).reshape(1, 8); active = list(range(8)); glob = [0] * 8; bpp = 96
else:
    xs, ys, ss, truths = M.make_global7(64, 100_000)
    ws = [1 / s for s in ss]
    tape = trace_model(M.model_global7, 7)
    pars = np.array([M.start_values(t) for t in truths]); pars[:, 4:] = M.start_values(M.GLOBAL7_TAUS)
    active = list(range(7)); glob = [0, 0, 0, 0, 1, 1, 1]; bpp = 88
pos = np.zeros(len(xs) + 1, dtype=np.int64)
for i, a in enumerate(xs):
    pos[i + 1] = pos[i] + len(a)
n = int(pos[-1])
X, Y, W = np.concatenate(xs), np.concatenate(ys), np.concatenate(ws)
res = {json.dumps(v, sort_keys=True): dict(fused=[], chi2=[], nostore=[]) for v in VARIANTS}
for r in range(rounds):
    for v in VARIANTS:
        os.environ.update(v)
        ctx = _lib.Context(0)
        ctx.set_placement_after(0)
        ctx.set_model(tape); ctx.set_data(X, Y, W, pos)
        jac, dim = ctx.jacobian_indices(active, glob)
        ctx.sweep(pars, active, jac, dim)
        ctx.time_kernel(5, 60)
        k = json.dumps(v, sort_keys=True)
        res[k]['fused'].append(ctx.time_kernel(5, 100))
        res[k]['chi2'].append(ctx.time_kernel(2, 100))
        ctx.set_keep_jacobian(0); ctx.sweep(pars, active, jac, dim); ctx.time_kernel(5, 40)
        res[k]['nostore'].append(ctx.time_kernel(5, 100))
        ctx.close()
for k, d in res.items():
    print('cfg%d %s' % (cfg, k), ' '.join('%s median %.4f min %.4f max %.4f (frac of 8 TB/s at median %.3f)' % (name, np.median(v), min(v), max(v), bpp * n / (np.median(v) * 1e-3) / 8e12) if name == 'fused'
                                         else '%s median %.4f' % (name, np.median(v)) for name, v in d.items()), flush=True)
