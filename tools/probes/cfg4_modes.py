#!/usr/bin/env python3
"""BASELINE config 4, ONE kind of launch per process (so that a counter pass attributes its numbers to it):
   python tools/probes/cfg4_modes.py <which> [reps]     which: 4 sweep (bisecting), 2 chi2, 3 omega (bisecting), 8 sweep replaying the
   recorded meshes, 9 omega replaying.  Prints the HIP-event average of the launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from gadfit_amd import _lib
from gadfit_amd.ad import trace_model
from tests import models as M
from tests.golden import goldens as G

which = int(sys.argv[1]); reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n = int(float(os.environ.get('CFG4_N', '1e6')))
a, b = 7.5, 0.8
xq = 0.05 + (10.0 - 0.05) * (np.arange(n) + 0.5) / n
from scipy.special import gammainc, gamma
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
import hashlib
ctx.time_kernel(which, 40)
# (the source key: sha1 of the generated translation unit -- bench.py refuses counts taken on another kernel)
print('which', which, 'avg_ms', ctx.time_kernel(which, reps), 'source_sha1', hashlib.sha1(ctx.model_source([0, 1]).encode()).hexdigest(),
      'ws_fast', os.environ.get('GADFIT_HIP_WS_FAST', 'default'))
ctx.close()
