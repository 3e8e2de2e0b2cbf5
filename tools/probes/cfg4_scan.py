#!/usr/bin/env python3
"""BASELINE config 4 kernels against N (tail / occupancy effects) -- python tools/probes/cfg4_scan.py N [descending]
(GADFIT_HIP_WS_FAST etc. from the environment).  Prints HIP-event averages: 4 sweep bisecting, 2 chi2, 3 omega bisecting."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from gadfit_amd import _lib
from gadfit_amd.ad import trace_model
from tests import models as M
from tests.golden import goldens as G
from scipy.special import gammainc, gamma

n = int(float(sys.argv[1]))
a, b = 7.5, 0.8
xq = 0.05 + (10.0 - 0.05) * (np.arange(n) + 0.5) / n
if len(sys.argv) > 2:
    xq = xq[::-1].copy()
fq = np.pi * 0.5 * b ** (-(a + 1) / 2) * gamma((a + 1) / 2) * gammainc((a + 1) / 2, b * xq * xq)
sq = 0.01 * (1 + np.abs(fq))
yq = fq + sq * M.normal(n, M.SEED)
t = trace_model(G.model_integral_single, 2); t.set_integration(rel_error=1e-10)
ctx = _lib.Context(0)
ctx.set_model(t)
ctx.set_data(xq, yq, 1.0 / sq, [0, n])
pars = np.array([[a * 1.05, b * 0.95]])
jac, dim = ctx.jacobian_indices([0, 1], [0, 0])
ctx.chi2(pars)
JTJ, JTr, chi2 = ctx.sweep(pars, [0, 1], jac, dim)
ctx.omega(pars, _lib.potr(JTJ + np.diag(np.diag(JTJ)), JTr))
out = []
for which in (4, 2, 3):
    ctx.time_kernel(which, 20)
    ms = ctx.time_kernel(which, 20)
    out.append('%d: %.3f ms = %.3f us/kpt' % (which, ms, 1e6 * ms / n))
print('N %d %s WS_FAST=%s ' % (n, 'desc' if len(sys.argv) > 2 else 'asc', os.environ.get('GADFIT_HIP_WS_FAST', '-')), '  '.join(out), flush=True)
ctx.close()
