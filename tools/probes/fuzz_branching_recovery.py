#!/usr/bin/env python3
"""a branching layout case of the Fortran fuzz through the PYTHON API, its variants explored at the start parameters only, so that
the device meets the other paths during the fit (as a Fortran program's capture does): device against the fully explored oracle.
   python tools/probes/fuzz_branching_recovery.py SEED"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests import fortran_fuzz as FZ
from gadfit_amd import _lib, tape as T
from oracle import binding as orc
seed = int(sys.argv[1])
c = FZ.make_layout_case(seed, branching=True)
root, nd = c['root'], c['nd']
rng = np.random.default_rng(88000 + seed)
sizes = [int(rng.integers(50, 300)) for _ in range(nd)]
xraw = [np.sort(rng.uniform(0.3, 1.6, size=n)) for n in sizes]
full = T.Variants(lambda p, x: 1.0 * root.fn(p, x), FZ.NP_)
for d in range(nd):
    cloud = [c['start'][d], c['truth'][d]] + [c['start'][d] * (1.0 + sc * rng.uniform(-1, 1, size=FZ.NP_)) for sc in (0.03, 0.1, 0.3) for _ in range(8)]
    for pp in cloud:
        full.explore(xraw[d], pp)
ys = []
for d in range(nd):
    f0 = orc.OracleProblem(full, [xraw[d]], [np.zeros_like(xraw[d])], [np.ones_like(xraw[d])], [c['truth'][d]], c['active'], [0] * FZ.NP_)
    y = -f0.sweep()[2]
    ys.append((np.abs(y) + 1.0) * (1.0 + 0.01 * rng.standard_normal(len(y))))
    rng.uniform(0.5, 2.0, size=len(y))
ws = [orc.init_weights(getattr(orc, c['mode']), y, np.ones_like(y)) for y in ys]
more = dict(c['more']); more.pop('use_ad', None)
kw = dict(lambda_=np.float32(c['lam']), max_iter=c['max_iter'])
if c['accth'] is not None: kw['accth'] = np.float32(c['accth'])
for k, v in more.items(): kw[k] = int(v) if isinstance(v, (bool, int)) else np.float32(v)
p = orc.OracleProblem(full, xraw, ys, ws, c['start'], c['active'], c['is_global'])
r0 = p.fit(**kw)
print('oracle : iterations', r0.iterations, 'chi2 %.15g' % r0.chi2, p.pars[:, c['active']].ravel())
start_only = T.Variants(lambda p, x: 1.0 * root.fn(p, x), FZ.NP_)
for d in range(nd):
    start_only.explore(xraw[d], c['start'][d])
print('variants: full', len(full.tapes), 'start only', len(start_only.tapes))
ctx = _lib.Context(0)
ctx.set_model(start_only)
X = np.concatenate(xraw); Y = np.concatenate(ys); W = np.concatenate(ws)
ctx.set_data(X, Y, W, list(np.concatenate([[0], np.cumsum(sizes)])))
out, r = ctx.fit(c['start'], c['active'], c['is_global'], **{k: (float(v) if isinstance(v, np.floating) else v) for k, v in kw.items()})
print('device : iterations', r.iterations, 'chi2 %.15g' % r.chi2, out[:, c['active']].ravel(), 'variants now', len(start_only.tapes))
ctx.close()
