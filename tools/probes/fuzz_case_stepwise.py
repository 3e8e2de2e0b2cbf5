#!/usr/bin/env python3
"""a single-dataset branching case of the Fortran fuzz, fitted as TWO gadf_fit calls of one iteration each (program and oracle):
separates what a pass computes from what a fit carries from one iteration to the next.   python tools/probes/fuzz_case_stepwise.py SEED N"""
import os, sys, tempfile, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests import fortran_fuzz as FZ
from tests import test_gpu_fortran_fuzz as T
from gadfit_amd import tape as TP
from oracle import binding as orc
seed, n_points = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(77000 + seed)
x = np.sort(rng.uniform(0.3, 1.6, size=n_points))
root, active, start, truth = FZ.make_branching_case(seed)
tape = TP.Variants(lambda p, x: 1.0 * root.fn(p, x), FZ.NP_)
for pp in [start, truth] + [start * (1.0 + 0.03 * rng.uniform(-1, 1, size=FZ.NP_)) for _ in range(6)]:
    tape.explore(x[:: max(1, n_points // 400)], pp)
f0 = orc.OracleProblem(tape, [x], [np.zeros_like(x)], [np.ones_like(x)], [truth], active, [0] * FZ.NP_)
JTJ0, _, res0, _ = f0.sweep(); y = -res0
keep = [k for q, k in enumerate(active) if JTJ0[q, q] > 1e-10 * np.max(np.diag(JTJ0))]
start = np.array([start[k] if k in keep else truth[k] for k in range(FZ.NP_)]); active = keep
y = y * (1.0 + 0.01 * rng.standard_normal(n_points))
work = tempfile.mkdtemp(prefix='fzstep')
data = os.path.join(work, 'data.txt')
with open(data, 'w') as fh:
    for a, b in zip(x, y):
        fh.write('%.17e %.17e\n' % (a, b))
x, y = np.loadtxt(data, unpack=True)
P = start.copy()
for it in (1, 2):
    p = orc.OracleProblem(tape, [x], [y], [np.ones_like(x)], [P], active, [0] * FZ.NP_)
    r0 = p.fit(lambda_=np.float32(1.0), max_iter=1)
    print('oracle fit', it, ': start', P[active], 'chi2_0 %.17g' % r0.chi2_0, 'JTres0', p.JTres0, 'JTJ0', p.JTJ0.ravel())
    P = p.pars[0].copy()
    print('oracle fit', it, 'chi2 %.15g' % r0.chi2, P[active])
src = FZ.fortran_source(root, active, start, 1.0, 1)
src = src.replace("  call gadf_fit(1.0, max_iter=1)\n", "  call gadf_fit(1.0, max_iter=1)\n  write(*, '(a, 5es22.14)') 'after fit 1: ', fitfuncs(1)%pars%val\n  call gadf_fit(1.0, max_iter=1)\n")
out = T._build_and_run(src, 'step_%d' % seed, [data], work)
print(out)
