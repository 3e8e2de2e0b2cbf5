#!/usr/bin/env python3
"""BASELINE config 4: does the ORDER in which workgroups meet expensive points matter?  The cost of a point grows with x (more
bisections); workgroups are dispatched in index order, so with ascending x the expensive ones start last and form the tail.
   python tools/probes/cfg4_order.py      prints sweep / chi2 / omega averages for ascending, descending and shuffled-by-block x."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from gadfit_amd import _lib
from gadfit_amd.ad import trace_model
from tests import models as M
from tests.golden import goldens as G
from scipy.special import gammainc, gamma

n = 1_000_000
a, b = 7.5, 0.8
xq = 0.05 + (10.0 - 0.05) * (np.arange(n) + 0.5) / n
fq = np.pi * 0.5 * b ** (-(a + 1) / 2) * gamma((a + 1) / 2) * gammainc((a + 1) / 2, b * xq * xq)
sq = 0.01 * (1 + np.abs(fq))
yq = fq + sq * M.normal(n, M.SEED)
t = trace_model(G.model_integral_single, 2); t.set_integration(rel_error=1e-10)
orders = {'ascending': np.arange(n), 'descending': np.arange(n)[::-1].copy()}
rng = np.random.default_rng(1)
blocks = rng.permutation(n // 500)
orders['blocks of 500 shuffled'] = (blocks[:, None] * 500 + np.arange(500)[None, :]).ravel()
for name, o in orders.items():
    ctx = _lib.Context(0)
    ctx.set_model(t)
    ctx.set_data(xq[o], yq[o], 1.0 / sq[o], [0, n])
    pars = np.array([[a * 1.05, b * 0.95]])
    jac, dim = ctx.jacobian_indices([0, 1], [0, 0])
    ctx.chi2(pars)
    JTJ, JTr, chi2 = ctx.sweep(pars, [0, 1], jac, dim)
    ctx.omega(pars, _lib.potr(JTJ + np.diag(np.diag(JTJ)), JTr))
    out = []
    for which in (4, 2, 3, 8, 9):
        ctx.time_kernel(which, 30)
        out.append('%d: %.3f ms' % (which, ctx.time_kernel(which, 30)))
    print('%-24s' % name, '  '.join(out), ' chi2 %.10g' % chi2, flush=True)
    ctx.close()
